"""usage: python tools/pmc_summary.py <counter_collection.csv> [kernel regex ...]
Per-kernel means of every counter in a rocprofv3 --pmc pass (kernels matching the given patterns)."""
import collections, csv, re, sys
pats = sys.argv[2:] or ['bin_faces_kernel2', 'raster_tile_kernel2', 'raster_backward', 'soft_select', 'soft_eval', 'soft_mask_backward_list', 'pv_forward', 'pv_backward']
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    for p in pats:
        if re.search(p, r['Kernel_Name']):
            agg[p][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for k in agg.values() for c in k})
print('kernel'.ljust(28) + ''.join(n[-18:].rjust(20) for n in names))
for k, cs in agg.items():
    print(k.ljust(28) + ''.join(f"{sum(cs[n]) / max(len(cs[n]), 1):20.0f}" for n in names))
