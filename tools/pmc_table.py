"""usage: python tools/pmc_table.py <counter_collection.csv> <out.txt> "<command line that produced it>"
Raw per-kernel table of every counter of one rocprofv3 --pmc pass (launches, mean and max of the raw counter value), kept
under profiles/ so that the figures DESIGN.md quotes can be recomputed."""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
res = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name']
    for junk in ('void ', '(anonymous namespace)::', 'at::native::', 'kamd::'):
        name = name.replace(junk, '')
    name = name[:100]
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    res[name] = (r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'], r['Workgroup_Size'], r['Grid_Size'])
counters = sorted({c for k in agg.values() for c in k})
with open(sys.argv[2], 'w') as fh:
    fh.write(f'# {sys.argv[3]}\n# 1x MI355X, own pass (no trace domains); raw counter units, mean over the launches of a kernel (max in brackets where it differs)\n')
    fh.write('# columns: launches | vgpr agpr sgpr lds wg grid(last) | ' + ' | '.join(counters) + ' | kernel\n')
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
        n = max(len(v) for v in cs.values())
        cells = []
        for c in counters:
            v = cs.get(c, [])
            if not v:
                cells.append('-')
                continue
            mean, mx = sum(v) / len(v), max(v)
            cells.append(f'{mean:.1f}' + (f' [{mx:.1f}]' if mx > mean * 1.01 else ''))
        fh.write(f'{n:5d} | ' + ' '.join(res[k]) + ' | ' + ' | '.join(cells) + f' | {k}\n')
print(open(sys.argv[2]).read()[:3000])
