"""usage: python tools/parse_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [source tag]
Calibrates FETCH_SIZE / WRITE_SIZE on launches of known size (see tools/pmc_traffic.py), then reports per-launch HBM
bytes of our kernels and the total of one DIB-R step (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts 16-B/lane streaming
reads at half their size; other widths uncalibrated -> we calibrate both counters on our own access patterns and say so)."""
import collections, csv, json, re, sys

STEPS = 3
# profile-table name (bench.py's "kernels" keys) -> pattern of the device symbol
OURS = {'bin_faces_kernel': r'bin_faces_kernel2', 'raster_tile_kernel': r'raster_tile_kernel2', 'raster_backward_kernel': r'raster_backward_(list_)?kernel',
        'soft_items_kernel': r'soft_items_kernel', 'soft_select_kernel': r'soft_select_kernel', 'soft_eval_kernel': r'soft_eval_kernel',
        'soft_mask_backward_list_kernel': r'soft_mask_backward_(list_kernel2|flat_kernel)', 'pv_forward_kernel': r'pv_forward_kernel',
        'pv_backward_kernel': r'pv_backward_kernel', 'fill_regions_kernel': r'fill_regions_kernel',
        'weighted_sum2_kernels': r'weighted_sum2_(partial|backward)_kernel'}   # (the forward's one-workgroup finish: 'other')


def rows(path, counter):
    out = [(int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])) for r in csv.DictReader(open(path))
           if r['Counter_Name'] == counter]
    return sorted(out)


SECTIONS = ['warmup', 'chamfer_step', 'chamfer_operator', 'voxelgrid_256', 'point_to_mesh_1Mx50k']   # tools/pmc_traffic.py, in order
#            ('warmup': the first call of every later section - allocator growth, code-object loads; not reported)


def sections(path, counter):
    """{section: {kernel: [values]}} of the launches behind each marker (a mask_iou call = forward + finish kernels)."""
    rs = rows(path, counter)
    out, cur, prev_marker = {}, None, False
    idx = -1
    for d, k, v in rs:
        is_marker = 'mask_iou' in k
        if is_marker:
            if not prev_marker:
                idx += 1
                cur = SECTIONS[idx] if idx < len(SECTIONS) else None
                if cur:
                    out[cur] = collections.defaultdict(list)
            prev_marker = True
            continue
        prev_marker = False
        if cur:
            out[cur][k].append(v)
    return out


def analyse(path, counter):
    rs = rows(path, counter)
    first_marker = min((d for d, k, v in rs if 'mask_iou' in k), default=None)
    if first_marker is not None:
        rs = [r for r in rs if r[0] < first_marker]
    last_fill = max((d for d, k, v in rs if 'fill_regions_kernel' in k), default=-1)
    # the steps begin with prepare_vertices: everything from the first pv_forward_kernel after the calibration on
    marker = min((d for d, k, v in rs if 'pv_forward_kernel' in k and d > last_fill), default=last_fill + 1) - 1
    clone = max((v for d, k, v in rs if '__amd_rocclr_copyBuffer' in k and d < last_fill), default=None)
    fill = [v for d, k, v in rs if 'fill_regions_kernel' in k]
    step = collections.defaultdict(list)
    for d, k, v in rs:
        if d > marker:
            step[k].append(v)
    return clone, (sum(fill) / len(fill) if fill else None), step


GiB = float(1 << 30)
clone_f, _, step_f = analyse(sys.argv[1], 'FETCH_SIZE')
clone_w, fill_w, step_w = analyse(sys.argv[2], 'WRITE_SIZE')
f_scale = GiB / clone_f if clone_f else None      # bytes per raw unit, streaming 16-B reads
w_scale = GiB / clone_w if clone_w else None
out = {'_source': sys.argv[4] if len(sys.argv) > 4 else 'tools/pmc_traffic.py',
       '_calibration': {'clone_1GiB_FETCH_SIZE_raw': clone_f, 'clone_1GiB_WRITE_SIZE_raw': clone_w,
                        'fill_regions_WRITE_SIZE_raw': fill_w, 'fill_regions_bytes': 8 * 1024 * 1024 * 30 * 13,
                        'bytes_per_FETCH_SIZE_unit': f_scale, 'bytes_per_WRITE_SIZE_unit': w_scale}}
claimed = set()
for name, pat in OURS.items():
    fk = [k for k in step_f if re.search(pat, k)]
    wk = [k for k in step_w if re.search(pat, k)]
    if not fk and not wk:
        continue
    claimed.update(fk + wk)
    fv = [v for k in fk for v in step_f[k]]
    wv = [v for k in wk for v in step_w[k]]
    fb = sum(fv) / len(fv) * f_scale if fv and f_scale else None
    wb = sum(wv) / len(wv) * w_scale if wv and w_scale else None
    out[name] = {'fetch_bytes': fb, 'write_bytes': wb, 'hbm_bytes': (fb or 0) + (wb or 0), 'launches_per_step': len(fv or wv) / STEPS}
# everything else dispatched inside the steps (torch's dot / fill / cat kernels, memsets)
other_f = sum(v for k, vs in step_f.items() if k not in claimed for v in vs) * (f_scale or 0) / STEPS
other_w = sum(v for k, vs in step_w.items() if k not in claimed for v in vs) * (w_scale or 0) / STEPS
ours = sum(v['hbm_bytes'] * v['launches_per_step'] for k, v in out.items() if not k.startswith('_') and k != 'fill_regions_kernel')
out['_step'] = {'our_kernels_hbm_bytes': ours, 'other_kernels_hbm_bytes': other_f + other_w,
                'hbm_bytes': ours + other_f + other_w,
                'other_kernels': sorted({re.sub(r'\(.*', '', k)[:80] for k in list(step_f) + list(step_w) if k not in claimed})}
# ---- chamfer / C5 sections: per-kernel bytes per launch and per call
sec_f, sec_w = sections(sys.argv[1], 'FETCH_SIZE'), sections(sys.argv[2], 'WRITE_SIZE')
calls = {'chamfer_step': STEPS, 'chamfer_operator': STEPS, 'voxelgrid_256': STEPS, 'point_to_mesh_1Mx50k': 1}
for sec in SECTIONS[1:]:
    if sec not in sec_f and sec not in sec_w:
        continue
    kernels = {}
    for k in sorted(set(sec_f.get(sec, {})) | set(sec_w.get(sec, {}))):
        fv, wv = sec_f.get(sec, {}).get(k, []), sec_w.get(sec, {}).get(k, [])
        name = re.sub(r'[(<].*', '', k.replace('(anonymous namespace)::', '').replace('void ', '').replace('kamd::', ''))[:70]
        e = kernels.setdefault(name, {'fetch_bytes_per_call': 0.0, 'write_bytes_per_call': 0.0, 'launches_per_call': 0.0})
        e['fetch_bytes_per_call'] += sum(fv) * (f_scale or 0) / calls[sec]
        e['write_bytes_per_call'] += sum(wv) * (w_scale or 0) / calls[sec]
        e['launches_per_call'] += max(len(fv), len(wv)) / calls[sec]
    out['_' + sec] = {'hbm_bytes_per_call': sum(e['fetch_bytes_per_call'] + e['write_bytes_per_call'] for e in kernels.values()),
                      'kernels': kernels}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
