"""Static instruction histogram of one kernel in a `hipcc -S --offload-device-only` listing.
usage: python tools/round4/isa_hist.py <file.s> <substring of the mangled kernel name> [--loops]
Counts per class (valu / valu_f64 / trans / salu / lds / vmem / smem / branch) for the whole function and, with --loops,
per basic block (label .. next label) so that the hot loop's body can be read off."""
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = None
for i, l in enumerate(src):
    if re.match(r'^_Z\S+:', l) and pat in l.split(':')[0]:
        start = i
        break
assert start is not None, 'kernel not found'
end = next(i for i in range(start, len(src)) if 's_endpgm' in src[i])
TRANS = ('v_rcp', 'v_rsq', 'v_sqrt', 'v_exp', 'v_log', 'v_sin', 'v_cos')
def cls(op):
    if op.startswith('v_'):
        if any(op.startswith(t) for t in TRANS): return 'trans'
        if 'f64' in op: return 'valu_f64'
        return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_load') or op.startswith('s_buffer'): return 'smem'
    if op.startswith(('s_cbranch', 's_branch')): return 'branch'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'): return 'wait'
    if op.startswith('s_'): return 'salu'
    return 'other'
tot = collections.Counter()
blocks = []
cur = ('entry', collections.Counter(), collections.Counter())
for l in src[start + 1:end + 1]:
    s = l.strip()
    if not s or s.startswith(';') or s.startswith('.'):
        if re.match(r'^\.LBB\d+_\d+:', s):
            blocks.append(cur)
            cur = (s.rstrip(':'), collections.Counter(), collections.Counter())
        continue
    op = s.split()[0]
    c = cls(op)
    tot[c] += 1
    cur[1][c] += 1
    cur[2][op] += 1
blocks.append(cur)
print('TOTAL', dict(tot), 'lines', end - start)
if '--loops' in sys.argv:
    for name, c, ops in blocks:
        n = sum(c.values())
        if n >= 12:
            top = ', '.join(f'{k}:{v}' for k, v in ops.most_common(8))
            print(f'{name:12s} n={n:4d} {dict(c)} | {top}')
