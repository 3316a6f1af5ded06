"""Register / LDS / scratch use of the kernels in a hipcc -S listing whose names contain a pattern."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    g = lambda k: re.search(r'\.amdhsa_' + k + r' (\S+)', body).group(1)
    print(name[:70], 'vgpr', g('next_free_vgpr'), 'sgpr', g('next_free_sgpr'), 'lds', g('group_segment_fixed_size'),
          'scratch', g('private_segment_fixed_size'))
