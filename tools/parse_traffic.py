"""usage: python tools/parse_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
Calibrates FETCH_SIZE / WRITE_SIZE on launches of known size (see tools/pmc_traffic.py), then reports per-launch HBM
bytes of our kernels (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts 16-B/lane streaming reads at half their size;
other widths uncalibrated -> we calibrate both counters on our own access patterns and say so)."""
import collections, csv, json, re, sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return agg


fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')


def find(agg, pat):
    return [(k, v) for k, v in agg.items() if re.search(pat, k)]


GiB = float(1 << 30)
# calibration 1: torch clone of 1 GiB (runs as __amd_rocclr_copyBuffer, 16 B/lane): reads 1 GiB, writes 1 GiB.
# calibration 2: fill_regions_kernel writes exactly B*H*W*K*13 bytes; soft_classify_kernel reads exactly 8 B/pixel.
clone_f = max((max(v) for k, v in find(fetch, '__amd_rocclr_copyBuffer')), default=None)
clone_w = max((max(v) for k, v in find(write, '__amd_rocclr_copyBuffer')), default=None)
fill_w = [sum(v) / len(v) for k, v in find(write, 'fill_regions_kernel')]
out = {'_calibration': {'clone_1GiB_FETCH_SIZE_raw': clone_f, 'clone_1GiB_WRITE_SIZE_raw': clone_w,
                        'fill_regions_WRITE_SIZE_raw': fill_w[0] if fill_w else None,
                        'fill_regions_bytes': 8 * 1024 * 1024 * 30 * 13}}
f_scale = GiB / clone_f if clone_f else None      # bytes per raw unit, streaming 16-B reads
w_scale = GiB / clone_w if clone_w else None
out['_calibration']['bytes_per_FETCH_SIZE_unit'] = f_scale
out['_calibration']['bytes_per_WRITE_SIZE_unit'] = w_scale
for name in ('raster_tile_kernel', 'raster_backward_kernel', 'bin_faces_raw_kernel', 'bin_faces_kernel', 'soft_classify_kernel',
             'soft_search_kernel', 'soft_mask_backward_list_kernel', 'fill_regions_kernel'):
    f = [sum(v) / len(v) for k, v in find(fetch, name + r'\b|' + name + '<')]
    w = [sum(v) / len(v) for k, v in find(write, name + r'\b|' + name + '<')]
    if not f and not w:
        continue
    fb = f[0] * f_scale if f and f_scale else None
    wb = w[0] * w_scale if w and w_scale else None
    out[name] = {'fetch_bytes': fb, 'write_bytes': wb, 'hbm_bytes': (fb or 0) + (wb or 0)}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
