"""usage: python tools/issue_table.py <pmc_SQ_waves_insts.txt> <pmc_SQ_lds_vmem.txt> <traffic.json> [out.txt]
What bounds each kernel, from the raw counter tables of one tools/round_profile.sh pass (tools/pmc_table.py wrote them):

  cycles      SQ_BUSY_CYCLES / 32    the launch's length in shader clocks (the counter is summed over the chip's 32 shader engines)
  valu        SQ_ACTIVE_INST_VALU * 4 / 1024 / cycles   the counter is in quad-clocks (one per wave64 vector instruction: a SIMD's issue
              cadence); MI355X has 256 CUs x 4 SIMDs.  An UPPER BOUND of the vector pipes' utilisation: a CDNA4 SIMD is 32 lanes wide
              and executes the instruction in 2 clocks.  Measured (profiles/r05z_soft_backward_diet_ab.txt): the soft mask's backward
              reads 1.08 here and takes the same time with 15 % fewer vector instructions.
  salu busy   SQ_INSTS_SALU / 256 / cycles              one scalar unit per CU, one instruction per clock
  lds busy    SQ_ACTIVE_INST_LDS * 4 / 256 / cycles      one LDS pipe per CU (quad-clocks as above) -- an upper bound
  occupancy   SQ_WAVE_CYCLES * 4 / (8192 * cycles)       resident wavefronts / the chip's 8192 slots
  waiting     SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES          share of a resident wavefront's time spent waiting for an instruction's operands
  hbm         calibrated FETCH + WRITE bytes per launch (traffic.json) / (cycles / 2.4 GHz) / 8 TB/s
No column near 1 = the launch is a chain of dependent round trips (latency), which is what every DIB-R kernel is."""
import json, re, sys


def table(path):
    cols, rows = None, {}
    for line in open(path):
        if line.startswith('# columns:'):
            cols = [c.strip() for c in line[len('# columns:'):].split('|')]
        elif not line.startswith('#') and cols:
            cells = [c.strip() for c in line.rstrip('\n').split('|')]
            name = cells[-1]
            vals = {}
            for c, v in zip(cols[2:-1], cells[2:-1]):
                vals[c] = float(v.split(' ')[0]) if v != '-' else None
            res = cells[1].split()
            vals['_vgpr'], vals['_sgpr'], vals['_lds'] = int(res[0]), int(res[2]), int(res[3])
            vals['_launches'] = int(cells[0])
            rows[name] = vals
    return rows


t1, t2 = table(sys.argv[1]), table(sys.argv[2])
traffic = json.load(open(sys.argv[3]))
out = open(sys.argv[4], 'w') if len(sys.argv) > 4 else sys.stdout
# the profile tables' kernel names -> the names of bench.py's kernel table / traffic.json
short = [('tl::bin_faces_kernel2<float, true, true>', 'bin_faces_kernel'), ('raster_tile_kernel2<float, true>', 'raster_tile_kernel'),
         ('soft_select_kernel<float, true>', 'soft_select_kernel'), ('soft_eval_kernel<float, true, true>', 'soft_eval_kernel'),
         ('soft_mask_backward_flat_kernel<float>', 'soft_mask_backward_list_kernel'), ('raster_backward_list_kernel<float', 'raster_backward_kernel'),
         ('pv_forward_kernel<float>', 'pv_forward_kernel'), ('pv_backward_kernel<float>', 'pv_backward_kernel'),
         ('weighted_sum2_partial_kernel<float>', None), ('weighted_sum2_backward_kernel<float>', None),
         ('sdg_build<float>', None), ('sdg_query<2, float, false, float>', None), ('ts_sweep_kernel<float, 256>', None),
         ('ts_hard_kernel<float>', None), ('vox_mark_kernel<float>', None), ('vox_clear_extent_kernel<float>', None)]
out.write('# ' + ' '.join(sys.argv) + '\n# ' + (traffic.get('_source') or '') + '\n')
out.write('# %-40s %9s %6s %6s %6s %6s %6s %6s  %s\n' % ('kernel', 'cycles', 'valu', 'salu', 'lds', 'occ', 'wait', 'hbm', 'vgpr/sgpr/lds'))
for prefix, tname in short:
    k1 = next((k for k in t1 if k.startswith(prefix)), None)
    k2 = next((k for k in t2 if k.startswith(prefix)), None)
    if k1 is None or k2 is None:
        continue
    a, b = t1[k1], t2[k2]
    cyc = a['SQ_BUSY_CYCLES'] / 32.0
    valu = b['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / cyc
    salu = a['SQ_INSTS_SALU'] / 256 / cyc
    lds = b['SQ_ACTIVE_INST_LDS'] * 4 / 256 / cyc
    occ = a['SQ_WAVE_CYCLES'] * 4 / (8192 * cyc)
    wait = a['SQ_WAIT_INST_ANY'] / a['SQ_WAVE_CYCLES']
    hbm = ''
    if tname and tname in traffic:
        hbm = '%.2f' % (traffic[tname]['hbm_bytes'] / (cyc / 2.4e9) / 8e12)
    out.write('  %-40s %9.0f %6.2f %6.2f %6.2f %6.2f %6.2f %6s  %d/%d/%d\n' % (re.sub(r'\(.*', '', k1)[:40], cyc, valu, salu, lds, occ, wait, hbm, a['_vgpr'], a['_sgpr'], a['_lds']))
