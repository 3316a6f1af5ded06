#!/usr/bin/env python
"""bench.py -- the hot path's headline benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Headline metric (BASELINE.json): Mpixels/s of DIB-R forward+backward at 1024x1024.  One "step" is one pass of
the hot path over one batch of synthetic camera views on every rank (config C4 of SURVEY.md 8(d)):
    prepare_vertices(shared vertices) -> dibr_rasterization(8 views/GPU of a 50 000-triangle geodesic sphere,
    D = 3 features, knum = 30, sigmainv = 7000, boxlen = 0.02) -> backward of (features*G1).sum() +
    (soft_mask*G2).sum() (kaolin_amd.metrics.render.weighted_sum: one fused pass each way; "torch_loss_variant" = the same
    loss as two torch dots) down to the shared vertices -> ONE all-reduce of the vertex gradient (N > 1).
Views are sharded over ranks (weak scaling: 8 views per GPU), no collective on the data path.  Inputs are
resident in HBM when the timed region starts.  The same run also times chamfer_distance fwd+bwd at
100k x 100k (config C3, one batch item per GPU) and reports it under "chamfer".

Rank 0 prints ONE JSON line.  "roofline" describes the kernel with the largest share of the DIB-R step, timed
live with HIP events on the launch stream inside the timed region (libkaolin_amd's kamd_profile_* hooks: the DIB-R kernels
are launched through hipExtLaunchKernelGGL, whose two events take the dispatch's own begin and end timestamps -- the interval
rocprofv3's kernel trace reports; the table of every kernel comes from a separate, fully instrumented pass of the same step
before it); "cpu_baseline" is the CPU
oracle (OpenMP) timed on a bounded sample on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import ctypes
import gc
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import kaolin_amd as kal  # noqa: E402
from kaolin_amd import _lib, distributed as D  # noqa: E402
from kaolin_amd.utils import testing as T  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is what a streaming copy reaches


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None,
                    help='ranks (one per GPU).  Under torch.distributed.run it must equal WORLD_SIZE; as a plain `python bench.py '
                         '--gpus N` with N > 1 the script launches its own N ranks (see relaunch_as_ranks)')
    ap.add_argument('--steps', type=int, default=100)   # (0.045 s of timed region: a single host hiccup is 0.3 % of it, not 2 %)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--views-per-gpu', type=int, default=8)
    ap.add_argument('--res', type=int, default=1024)
    ap.add_argument('--sphere-frequency', type=int, default=50, help='20*f^2 triangles (50 -> 50 000)')
    ap.add_argument('--chamfer-points', type=int, default=100000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-chamfer', action='store_true')
    ap.add_argument('--no-cpu-extras', action='store_true',
                    help='skip cpu_baseline.other_paths: the C oracle / the torch oracles of chamfer and the torch oracle of the '
                         'rasterizer (BASELINE.md section 3), each on a bounded sample, in a child process with a hard time limit')
    ap.add_argument('--cpu-extras-only', action='store_true', help=argparse.SUPPRESS)   # the child process of the line above
    ap.add_argument('--no-contract-ops', action='store_true',
                    help='skip the timing of the four reference-contract (K-buffer) operators at C4')
    ap.add_argument('--no-c5', action='store_true', help='skip the voxelgrid / point-to-mesh extras (config C5)')
    ap.add_argument('--quick', action='store_true',
                    help='development probe: the headline step and its kernel table only (no variants, graph replay, contract '
                         'operators, chamfer, C5, CPU baseline)')
    ap.add_argument('--scene', choices=['sphere', 'knot', 'knot_shuffled'], default='sphere',
                    help='the mesh of the headline step: sphere = config C4 (BASELINE.json; the default and the only one `value` may be '
                         'quoted on), knot = the non-convex ~49k-triangle scene of kaolin_amd.utils.testing.knot_mesh (depth complexity up '
                         'to 8-11, image-sized triangles, geometry leaving the image); the default run also times the knot scene and reports '
                         'it under "scene_variants"')
    ap.add_argument('--no-scene-variants', action='store_true', help='skip the timing of the other scene')
    ap.add_argument('--dump-vertex-grad', default=None, metavar='FILE',
                    help='tests: after the timed steps run one more step and save rank 0\'s (all-reduced) vertex gradient with torch.save')
    ap.add_argument('--look-at', type=float, nargs=3, default=[0., 0., 0.],
                    help='point the cameras look at (default: the mesh centre; e.g. 0.35 -0.3 0 renders the object off-centre)')
    return ap.parse_args()


def algorithmic_bytes(kernel, B, P, F, Fv, D, K, esz=4, P_cov=None):
    """Contract bytes per launch: every operator input the kernel consumes read once, every operator
    output it produces written once (intermediate records / lists are NOT counted: they are this design's own traffic).
    `P_cov` (pixels of the 16 x 16 tiles that hold a covered pixel, per view; None = every pixel): the fused operator's
    backward walks the forward's list of those tiles and never reads the others -- charging it every pixel put it at 0.97 of
    HBM peak in round 3's line, above what any kernel reaches."""
    Pb = P if P_cov is None else P_cov
    per = {
        # P (20 + 4D) out -- face_idx (i64), 3 weights, D features -- + F' (52 + 12D) in -- the front faces'
        # 13 scalars + 3 D feature scalars: 32 B/pixel + 88 B/front face at D = 3.  (The fused operator's launch also writes
        # the soft mask of the pixels it settles, 4 B/pixel, and skips the 12 B/pixel of background weights: see
        # roofline.bytes_note / roofline.launch_bytes)
        'raster_tile_kernel': P * (8 + 3 * esz + D * esz) + Fv * (13 * esz + 3 * D * esz),
        # K2 in the fused operator (static features: no feature gradient): face_idx, weights and the upstream gradient of the
        # covered tiles in; per front face 6 image coordinates in, 6 gradient values out
        'raster_backward_kernel': Pb * (8 + 3 * esz + D * esz) + Fv * (6 * esz * 2),
        'fill_regions_kernel': P * K * (esz + 8 + 1),
        # select reads face_idx of the uncovered pixels' tiles and the faces' 6 coordinates + 4 box scalars
        'soft_select_kernel': P * 8 + F * 10 * esz,
        'soft_mask_backward_kernel': P * (8 + 2 * esz) + F * 6 * esz * 2,
        'soft_mask_backward_list_kernel': P * (2 * esz) + F * 6 * esz * 2,
        'bin_faces_kernel': F * (13 * esz),
    }
    return B * per[kernel] if kernel in per else None


STREAM_COPY_GBS = 6300.0   # what a streaming copy reaches on MI355X (HBM_PEAK_GBS note): no kernel moves its bytes faster


# kernels of the fused DIB-R operator that MAY share the GPU with a concurrent launch in the timed region: the backward's pair.  Since
# round 2 they run one after the other on the caller's stream in the product build (the side stream is an experiment-build knob,
# KAMD_BWD_SIDE_STREAM=1); ties between near-equal shares are still resolved away from them
OVERLAPPED = {'raster_backward_kernel', 'soft_mask_backward_list_kernel'}


def pick_dominant(kernels):
    """Name of the kernel the roofline line describes: the largest share of the instrumented step among the kernels with
    an algorithmic byte count; shares within 5 % of each other (raster_tile and soft_search tie) are resolved towards a
    kernel that runs alone on the stream in the timed region, so that its event pair times that kernel only."""
    ranked = sorted((k for k in kernels if kernels[k]['algorithmic_GBps'] is not None),
                    key=lambda k: -kernels[k]['share_of_instrumented_step'])
    if not ranked:
        return None
    top = kernels[ranked[0]]['share_of_instrumented_step']
    near = [k for k in ranked if kernels[k]['share_of_instrumented_step'] >= 0.95 * top]
    return next((k for k in near if k not in OVERLAPPED), ranked[0])


def cpu_reference_extras(n_points=100000, slice_rows=4096, raster_res=32, sphere_frequency=50, seed=0):
    """The other CPU readings BASELINE.md section 3 plans beside the DIB-R oracle figure, each on a bounded sample (CPU
    only; a few seconds together):
      * chamfer, restated kernel: the C oracle of K5 (OpenMP) on `slice_rows` query rows against all `n_points` targets,
        both directions -- the brute-force cost is linear in the rows, so pairs/s of the slice is pairs/s of the item;
      * chamfer, torch oracle: the dense `_sided_distance` formulation of the reference's tests on a 256-row chunk per direction;
      * rasterizer, torch oracle: `_naive_deftet_sparse_render(knum=1)` (what the reference's rasterizer tests are
        pinned to) on raster_res^2 pixels of view 0 of the bench's mesh."""
    import oracle
    oracle.build()
    threads = min(os.cpu_count() or 1, 16)   # (the dense formulation's temporaries scale badly beyond a few dozen threads)
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    p1, p2 = torch.rand((1, n_points, 3), generator=g), torch.rand((1, n_points, 3), generator=g)
    rows = min(slice_rows, n_points)
    t0 = time.perf_counter()
    d12, i12 = oracle.sided_distance_forward(p1[:, :rows], p2, omp=True)
    d21, i21 = oracle.sided_distance_forward(p2[:, :rows], p1, omp=True)
    oracle.sided_distance_backward(torch.ones_like(d12), p1[:, :rows], p2, i12)
    oracle.sided_distance_backward(torch.ones_like(d21), p2[:, :rows], p1, i21)
    dt_c = time.perf_counter() - t0
    out = {'chamfer_restated_kernel': {
        'value': round(2.0 * rows * n_points / dt_c / 1e6, 1), 'unit': 'Mpoint-pairs/s', 'cores': oracle.num_threads(True), 'kind': 'port',
        'sample': f'{rows} query rows x {n_points} targets, both directions fwd + bwd, C oracle (OpenMP over rows), {dt_c:.2f} s'}}
    chunk = min(64, rows)     # (64 x 100 000 x 3 floats = 77 MB per temporary)
    t0 = time.perf_counter()
    done, worst = 0, 0.0
    with torch.no_grad():
        while done + chunk <= rows and time.perf_counter() - t0 < 3.0:       # 64-row chunks for about three seconds
            t12 = kal.metrics.pointcloud._sided_distance(p1[:, done:done + chunk], p2)
            t21 = kal.metrics.pointcloud._sided_distance(p2[:, done:done + chunk], p1)
            for t, d in ((t12, d12), (t21, d21)):
                ref = d[:, done:done + chunk]
                worst = max(worst, float(((t - ref).abs() / ref.clamp(min=1e-30)).max()))
            done += chunk
    dt_t = time.perf_counter() - t0
    out['chamfer_torch_oracle'] = {
        'value': round(2.0 * done * n_points / dt_t / 1e6, 1), 'unit': 'Mpoint-pairs/s', 'cores': threads, 'kind': 'port',
        'max_rel_diff_vs_restated_kernel': worst,
        'sample': f'_sided_distance (the dense torch formulation the reference\'s tests use as their oracle, restated; forward values '
                  f'only) on {done // chunk} chunks of {chunk} x {n_points} per direction, {dt_t:.2f} s'}
    # rasterizer: view 0 of the bench's scene on a coarse pixel grid (the cost per pixel does not depend on the resolution)
    verts, faces = T.geodesic_sphere(sphere_frequency)
    cams = T.fibonacci_cameras(8, 2.5)[:1]
    rot, trans = kal.render.camera.generate_rotate_translate_matrices(cams, torch.zeros((1, 3)), torch.tensor([[0., 1., 0.]]))
    proj = kal.render.camera.generate_perspective_projection(math.pi / 4)
    with torch.no_grad():
        fv_cam, fv_img, _ = kal.render.mesh.prepare_vertices(verts.float().unsqueeze(0), faces, proj, camera_rot=rot, camera_trans=trans)
    fz = fv_cam[..., 2].contiguous()
    F = faces.shape[0]
    feats = torch.rand((1, F, 3, 3), generator=g)
    r = raster_res
    xs = (2. * torch.arange(r, dtype=torch.float) + 1. - r) / r
    ys = (r - 2. * torch.arange(r, dtype=torch.float) - 1.) / r
    pix = torch.stack([xs.unsqueeze(0).expand(r, r), ys.unsqueeze(1).expand(r, r)], dim=-1).reshape(1, r * r, 2)
    rng = torch.tensor([[[float(fz.min()) - 1e-2, float(fz.max()) + 1e-2]]]).expand(1, r * r, 2).contiguous()
    a_img = fv_img.clone().requires_grad_()
    t0 = time.perf_counter()
    img, idx = kal.render.mesh.deftet._naive_deftet_sparse_render(pix, rng, fz, a_img, feats, 1)
    img.sum().backward()
    dt_r = time.perf_counter() - t0
    ref_idx = oracle.rasterize(r, r, fz, fv_img, feats, omp=True)[1]
    out['rasterize_torch_oracle'] = {
        'value': round(r * r / dt_r / 1e6, 6), 'unit': 'Mpixels/s', 'cores': threads, 'kind': 'port',
        'face_idx_equals_restated_kernel': bool(torch.equal(idx[..., 0].reshape(1, r, r), ref_idx)),
        'sample': f'_naive_deftet_sparse_render(knum=1) fwd + autograd bwd (rasterizer only, no soft mask) on {r}x{r} pixels of one view of '
                  f'the {F}-triangle mesh, {dt_r:.2f} s'}
    return out


def time_contract_operators(verts, faces, proj, rot, trans, feats3, G1, G2, H, W, reps):
    """ms per call of packed_rasterize_forward_cuda / rasterize_backward_cuda / dibr_soft_mask_forward_cuda /
    dibr_soft_mask_backward_cuda on the bench's 8 views (reference: kaolin/csrc/render/mesh/rasterization.cpp:49-168,
    dibr_soft_mask.cpp:48-183), with the operator-contract bytes each one moves."""
    m = kal._C.render.mesh
    V = rot.shape[0]
    F = faces.shape[0]
    with torch.no_grad():
        fv_cam, fv_img, normals = kal.render.mesh.prepare_vertices(verts.detach().unsqueeze(0).expand(V, -1, -1), faces, proj,
                                                                   camera_rot=rot, camera_trans=trans)
        fz, nz = fv_cam[..., 2].contiguous(), normals[..., 2]
        # the reference's glue (rasterization.py:292-327, dibr.py:31-39), outside the timed calls
        valid = nz >= 0
        mesh_of, faces_of = torch.where(valid)
        packed_img = (fv_img[mesh_of, faces_of] * 1000.).contiguous()
        packed_z, packed_feat = fz[mesh_of, faces_of].contiguous(), feats3[mesh_of, faces_of].contiguous()
        first_idx = torch.zeros(V + 1, dtype=torch.long, device=fz.device)
        first_idx[1:] = torch.cumsum(valid.sum(dim=1), dim=0)
        bboxes = torch.cat((packed_img.min(dim=1)[0], packed_img.max(dim=1)[0]), dim=1).contiguous()
        scaled = (fv_img * 1000.).contiguous()
        large = torch.cat([scaled.min(dim=-2)[0] - 20., scaled.max(dim=-2)[0] + 20.], dim=-1).contiguous()
    Fp = int(first_idx[-1])
    P = V * H * W

    def ev_time(fn):
        """median of `reps` event-bracketed calls after three warm-up calls; the previous call's outputs are dropped BEFORE the
        next call allocates its own (dibr_soft_mask_forward_cuda returns 3.3 GB of K-buffers at C4: with two sets alive the
        timed call paid for allocator growth -- 6.5 ms on the driver's box against 0.8-0.96 ms everywhere else in round 3)"""
        out = None
        for _ in range(3):
            del out
            out = fn()
        torch.cuda.synchronize()
        times = []
        for _ in range(reps):
            del out
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn()
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1))
        times.sort()
        return times[len(times) // 2], out

    t_k1, (interp, sel, wts) = ev_time(lambda: m.packed_rasterize_forward_cuda(H, W, packed_z, packed_img, bboxes, packed_feat,
                                                                              first_idx, 1000., 1e-8))
    lookup = (sel + first_idx[:-1].reshape(-1, 1, 1)).reshape(-1)
    face_idx = faces_of[lookup.clamp_(min=0, max=max(Fp - 1, 0))].reshape(sel.shape).contiguous()
    face_idx[sel == -1] = -1
    t_k2, _ = ev_time(lambda: m.rasterize_backward_cuda(G1, interp, face_idx, wts, fv_img, feats3, 1e-8))
    t_k3, (soft, prob, idx, typ) = ev_time(lambda: m.dibr_soft_mask_forward_cuda(scaled, large, face_idx, 7000., 30, 1000.))
    t_k4, _ = ev_time(lambda: m.dibr_soft_mask_backward_cuda(G2, soft, face_idx, prob, idx, typ, scaled, 7000., 1000.))
    D_, K = 3, 30
    by = {'packed_rasterize_forward_cuda': P * (20 + 4 * D_) + Fp * (52 + 12 * D_),
          'rasterize_backward_cuda': P * (20 + 4 * D_) + V * F * (48 + 24 * D_),
          'dibr_soft_mask_forward_cuda': P * (12 + 13 * K) + V * F * 40,
          'dibr_soft_mask_backward_cuda': P * (16 + 13 * K) + V * F * 48}
    ms = {'packed_rasterize_forward_cuda': t_k1, 'rasterize_backward_cuda': t_k2, 'dibr_soft_mask_forward_cuda': t_k3,
          'dibr_soft_mask_backward_cuda': t_k4}
    ops = {k: {'ms': round(ms[k], 4), 'contract_bytes': by[k], 'GBps': round(by[k] / (ms[k] * 1e-3) / 1e9, 1),
               'frac_of_8TBps': round(by[k] / (ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} for k in ms}
    tot_ms, tot_b = sum(ms.values()), sum(by.values())
    del prob, idx, typ
    return {'views': V, 'ops': ops, 'sum_ms': round(tot_ms, 4), 'contract_bytes': tot_b,
            'GBps': round(tot_b / (tot_ms * 1e-3) / 1e9, 1), 'frac_of_8TBps': round(tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            'Mpixels_per_s': round(P / (tot_ms * 1e-3) / 1e6, 1),
            'note': 'the four reference-contract operators called one after the other as the reference\'s autograd Functions call '
                    'them (K-buffers materialised: 390 B/pixel written by K3, read by K4); operator calls only, the torch glue '
                    'around them is outside the events; ms = median of the event-bracketed calls after 3 warm-up calls'}


def relaunch_as_ranks(n):
    """`python bench.py --gpus N` (N > 1) without a launcher: run the same command line as N ranks under
    torch.distributed.run on this node (rendezvous on 127.0.0.1, a free port) and exit with its status; rank 0 of that job prints
    the JSON line.  With fewer visible GPUs than ranks the ranks share devices round robin -- RCCL refuses two ranks on one device,
    so gloo carries the (300 KB per step) collectives then, and the line says so (`distributed.ranks_per_gpu`)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if torch.cuda.device_count() < n:
        env.setdefault('KAMD_DIST_BACKEND', 'gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.quick:
        args.no_cpu_baseline = args.no_chamfer = args.no_c5 = args.no_contract_ops = True
    if args.cpu_extras_only:
        print(json.dumps(cpu_reference_extras(args.chamfer_points, sphere_frequency=args.sphere_frequency)))
        return
    launched = 'WORLD_SIZE' in os.environ or 'RANK' in os.environ     # a launcher (torch.distributed.run) made this process a rank
    if args.gpus is not None and args.gpus > 1 and not launched:
        relaunch_as_ranks(args.gpus)
    D.init_from_env()
    rank, world = D.rank(), D.world_size()
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the job has {world} rank(s) (WORLD_SIZE={os.environ.get("WORLD_SIZE")}): '
                         'launch one rank per GPU, or run plain `python bench.py --gpus N`')
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP operators have no CPU fallback)'
    dev = torch.device('cuda', torch.cuda.current_device())
    lib = _lib.load()
    H = W = args.res
    V = args.views_per_gpu
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')   # HBM bytes per launch from the PMC passes (tools/pmc_traffic.py)

    # ---------------- synthetic scene (config C4): shared mesh, this rank's slice of the camera ring
    def build_scene(name):
        verts, faces = T.scene_mesh(name, args.sphere_frequency)
        sc = {'name': name, 'verts': verts.float().to(dev).requires_grad_(), 'faces': faces.to(dev), 'F': faces.shape[0]}
        cams = D.shard_views(T.fibonacci_cameras(V * world, 2.5)).to(dev)     # this rank's views of the shared mesh
        look_at = torch.tensor([args.look_at], device=dev, dtype=torch.float).repeat(V, 1)
        up = torch.tensor([[0., 1., 0.]], device=dev).repeat(V, 1)
        sc['rot'], sc['trans'] = kal.render.camera.generate_rotate_translate_matrices(cams, look_at, up)
        g = torch.Generator().manual_seed(0)
        uv = torch.rand((1, sc['F'], 3, 2), generator=g).to(dev).expand(V, -1, -1, -1).contiguous()
        ones = torch.ones((V, sc['F'], 3, 1), device=dev)
        sc['feats3'] = torch.cat([uv, ones], dim=-1).contiguous()   # D = 3 (uv + mask channel, as in the tutorial), static input
        # The vertex gradient is shared by every view: its all-reduce is posted from autograd's accumulate hook and awaited
        # before the step ends; with one process this is a no-op.
        sc['reducer'] = D.SharedGradientReducer([sc['verts']])
        return sc

    proj = kal.render.camera.generate_perspective_projection(math.pi / 4).to(dev)
    # the loss weights of a view depend on its GLOBAL index only: N ranks of 8 views compute exactly the loss one process would
    # compute on the same 8 N views (tests/test_distributed.py compares the all-reduced vertex gradient of the two)
    first_view = D.shard_range(V * world)[0]
    G1 = torch.stack([torch.rand((H, W, 3), generator=torch.Generator().manual_seed(1000 + first_view + v)) for v in range(V)]).to(dev)
    G2 = torch.stack([torch.rand((H, W), generator=torch.Generator().manual_seed(5000 + first_view + v)) for v in range(V)]).to(dev)
    G1f, G2f = G1.reshape(-1), G2.reshape(-1)
    target_mask = (G2 > 0.5).float()
    scene = build_scene(args.scene)
    verts, faces, F, rot, trans, feats3, reducer = (scene[k] for k in ('verts', 'faces', 'F', 'rot', 'trans', 'feats3', 'reducer'))
    feats3_grad = feats3.clone().requires_grad_()          # variant: learnable per-face features (a texture atlas per view)

    def make_step(sc, features, tutorial_loss=False, torch_loss=False):
        verts_, faces_, rot_, trans_, reducer_ = sc['verts'], sc['faces'], sc['rot'], sc['trans'], sc['reducer']

        def step():
            verts_.grad = None
            features.grad = None
            fv_cam, fv_img, normals = kal.render.mesh.prepare_vertices(
                verts_.unsqueeze(0).expand(V, -1, -1), faces_, proj, camera_rot=rot_, camera_trans=trans_)
            feat, soft, face_idx = kal.render.mesh.dibr_rasterization(
                H, W, fv_cam[..., 2], fv_img, features, normals[..., 2])
            if tutorial_loss:
                # the DIB-R tutorial's objective: L1 image loss + silhouette IoU (kaolin.metrics.render.mask_iou, fused here)
                loss = torch.mean(torch.abs(feat - G1)) + kal.metrics.render.mask_iou(soft, target_mask)
            elif torch_loss:
                # the linear loss in plain torch, written as two dot products (rocBLAS: one pass each way per dot)
                loss = torch.dot(feat.reshape(-1), G1f) + torch.dot(soft.reshape(-1), G2f)
            else:
                # (features * G1).sum() + (soft_mask * G2).sum(): one fused pass over both G-buffers each way
                loss = kal.metrics.render.weighted_sum(feat, G1, soft, G2)
            loss.backward()
            reducer_.wait()
            return face_idx
        return step
    dibr_step = make_step(scene, feats3)
    dibr_step_tutorial = make_step(scene, feats3, tutorial_loss=True)
    dibr_step_torch_loss = make_step(scene, feats3, torch_loss=True)
    dibr_step_feature_grad = make_step(scene, feats3_grad)

    def per_step_ms(fn, steps):
        """Duration of every step of one more pass (events between steps on the launch stream, no host sync inside)."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        torch.cuda.synchronize()
        ev[0].record()
        for i in range(steps):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        d = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
        return {'median': round(d[len(d) // 2], 4), 'min': round(d[0], 4), 'p90': round(d[min(len(d) - 1, int(0.9 * len(d)))], 4),
                'max': round(d[-1], 4), 'steps': steps}

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        # No collector pause inside the timed region (the step allocates a few hundred Python objects) -- and NO gc.collect() in front
        # of it: a full collection here hands the caching allocator every buffer the last steps' garbage still held, the first timed
        # step then takes the host 0.83 instead of 0.34 ms to enqueue and the following ones run on different blocks, 3 % slower
        # (K = 20 after W = 5: 0.454 ms per step with the collection, 0.418 without; profiles/r06zs_timed_region.txt).  The warm-up
        # steps have just put the allocator into its steady state: the timed steps should start from it.
        gc.disable()
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        timed.enqueue_ms = (time.perf_counter() - t0) / steps * 1e3   # host time to enqueue a step (GPU still running)
        torch.cuda.synchronize()
        D.barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        if D.is_distributed():
            t = torch.tensor([dt], device=dev, dtype=torch.double)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def host_enqueue_idle_ms(fn, steps=20):
        """Host time to enqueue ONE step with the GPU idle (a synchronize between steps, outside the clock): what the Python / torch /
        HIP-runtime side of a step costs by itself.  `host_enqueue_ms_per_step` -- the enqueue time of the timed region's K steps
        back to back -- also contains the waits of a host that runs ahead of a slower GPU until the runtime's queue is full: on a
        GPU-bound step it converges to the GPU's time per step and says nothing about the host."""
        torch.cuda.synchronize()
        tot = 0.0
        for _ in range(steps):
            t0 = time.perf_counter()
            fn()
            tot += time.perf_counter() - t0
            torch.cuda.synchronize()
        return tot / steps * 1e3

    def covered_tile_pixels(face_idx):
        """pixels of the 16 x 16 tiles that hold a covered pixel, per view on average (what the fused backward's tile walk reads)"""
        c = (face_idx >= 0)
        Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
        c = torch.nn.functional.pad(c, (0, Wp - W, 0, Hp - H))
        tiles = c.reshape(V, Hp // 16, 16, Wp // 16, 16).any(dim=4).any(dim=2)
        return float(tiles.float().sum()) * 256.0 / V

    def kernel_table(step, sc, steps):
        """One fully instrumented pass (every launch of the library timed with HIP events: the DIB-R kernels by their own begin / end
        timestamps, multi-launch operators by two records around them): per-kernel average durations."""
        lib.kamd_profile_reset()
        lib.kamd_profile_select(-1)
        lib.kamd_profile_enable(1)
        inst_dt = timed(step, steps, 0)
        lib.kamd_profile_enable(0)
        prof = _lib.kernel_profile(reset=True)
        inst_ms = inst_dt / steps * 1e3
        face_idx = step()
        front = sc['front_faces']
        p_cov = covered_tile_pixels(face_idx)
        table, over = {}, []
        for name, (ms, n) in prof.items():
            avg_us = ms / n * 1e3
            ab = algorithmic_bytes(name, V, H * W, sc['F'], front, 3, 30, P_cov=p_cov)
            gbps = None if ab is None else round(ab / (avg_us * 1e-6) / 1e9, 1)
            if gbps is not None and gbps > STREAM_COPY_GBS:
                over.append(name)      # (a byte model that charges a kernel bytes it does not move: reported, never silently kept)
            table[name] = {'avg_us': round(avg_us, 2), 'launches_per_step': round(n / steps, 2),
                           'share_of_instrumented_step': round(ms / steps / inst_ms, 4), 'algorithmic_GBps': gbps}
        return table, inst_ms, face_idx, p_cov, over

    def soft_work_units(sc, face_idx, boxlen=0.02, knum=30):
        """What the soft-mask pass of a scene has to do, counted from the scene (per view, averages): `soft_pixels` = uncovered pixels
        inside at least one face's enlarged box (the pixels the reference's kernel computes distances for), `soft_items` = the
        16 x 4 sub-tiles holding one, `soft_pairs` = (pixel,
        face) pairs with the pixel inside the face's enlarged box (every one is a point-triangle distance in the reference: its
        kernel's work unit), `soft_hits` = the pairs that make it into the knum-deep buffers.  Box counts per pixel through a 2-D
        difference array.  (Pixel centres as dibr_soft_mask_cuda.cu:75-76: x = (2 px + 1 - W) / W, y = (H - 1 - 2 py) / H.)"""
        with torch.no_grad():
            _, fv_img, _ = kal.render.mesh.prepare_vertices(sc['verts'].detach().unsqueeze(0).expand(V, -1, -1), sc['faces'], proj,
                                                           camera_rot=sc['rot'], camera_trans=sc['trans'])
            x, y = fv_img[..., 0].double(), fv_img[..., 1].double()
            x0, x1 = x.min(-1).values - boxlen, x.max(-1).values + boxlen
            y0, y1 = y.min(-1).values - boxlen, y.max(-1).values + boxlen
            c0 = torch.ceil((x0 * W + W - 1) / 2).clamp(0, W).long()
            c1 = torch.floor((x1 * W + W - 1) / 2).clamp(-1, W - 1).long() + 1
            r0 = torch.ceil((H - 1 - y1 * H) / 2).clamp(0, H).long()
            r1 = torch.floor((H - 1 - y0 * H) / 2).clamp(-1, H - 1).long() + 1
            ok = (c1 > c0) & (r1 > r0)
            diff = torch.zeros((V, H + 1, W + 1), dtype=torch.int32, device=dev)
            vi = torch.arange(V, device=dev).unsqueeze(1).expand_as(c0)[ok]
            one = torch.ones_like(vi, dtype=torch.int32)
            for rr, cc, sgn in ((r0, c0, 1), (r0, c1, -1), (r1, c0, -1), (r1, c1, 1)):
                diff.index_put_((vi, rr[ok], cc[ok]), one * sgn, accumulate=True)
            cnt = diff.cumsum(1).cumsum(2)[:, :H, :W]
            unc = face_idx < 0
            per_pixel = cnt[unc]
            out = {'soft_pixels': float((per_pixel > 0).sum()) / V, 'soft_pairs': float(per_pixel.sum()) / V,
                   'soft_hits': float(per_pixel.clamp(max=knum).sum()) / V}
            if H % 4 == 0 and W % 16 == 0:
                # `soft_items`: the 16 x 4-pixel sub-tiles that hold such a pixel -- the search kernels' work items (a wavefront each)
                reach = (cnt > 0) & unc
                out['soft_items'] = float(reach.reshape(V, H // 4, 4, W // 16, 16).any(dim=4).any(dim=2).sum()) / V
            return out

    # the unit of work each kernel's duration scales with (scene_variants: a kernel's duration ratio between two scenes is judged
    # against the ratio of these, not against 1)
    KERNEL_WORK_UNIT = {'raster_tile_kernel': 'covered_tile_pixels', 'raster_backward_kernel': 'covered_tile_pixels',
                        # (select and eval run a wavefront per item, whatever the item holds: a scene of sparse items -- image-sized faces,
                        # whose enlarged boxes put large uncovered areas within reach of one or two faces -- costs them per item)
                        'soft_select_kernel': 'soft_items', 'soft_eval_kernel': 'soft_items', 'soft_mask_backward_list_kernel': 'soft_hits',
                        'bin_faces_kernel': 'faces', 'pv_forward_kernel': 'faces', 'pv_backward_kernel': 'faces',
                        'weighted_sum2_kernels': 'pixels'}

    def work_units(sc, face_idx, p_cov_):
        u = {'faces': float(sc['F']), 'pixels': float(H * W), 'covered_pixels': float((face_idx >= 0).float().sum()) / V,
             'covered_tile_pixels': float(p_cov_)}
        u.update(soft_work_units(sc, face_idx))
        return {k: round(v, 1) for k, v in u.items()}

    def front_faces(sc):
        """front-facing faces per view on average (what K1 / K2 read): counted from the scene, not assumed"""
        with torch.no_grad():
            _, _, normals = kal.render.mesh.prepare_vertices(sc['verts'].detach().unsqueeze(0).expand(V, -1, -1), sc['faces'], proj,
                                                            camera_rot=sc['rot'], camera_trans=sc['trans'])
        return float((normals[..., 2] >= 0).float().sum()) / V

    # ---------------- DIB-R.  Timing a launch costs stream time (a profiled dispatch, or two event records), and timing every
    # kernel also keeps the operator's two concurrent launches (side stream) on one stream.  So: (1) an instrumented
    # pass outside the timed region gives the per-kernel table and names the dominant kernel; (2) the timed region runs
    # the step as users run it, with HIP events around the dominant kernel only (the roofline line's duration).
    scene['front_faces'] = front_faces(scene)
    Fv = scene['front_faces']
    for _ in range(args.warmup):
        face_idx = dibr_step()
    kernel_ids = {lib.kamd_profile_kernel_name(k).decode(): k for k in range(lib.kamd_profile_num_kernels())}
    kernels, inst_ms_per_step, face_idx, p_cov, over_peak = kernel_table(dibr_step, scene, args.steps)
    dom = pick_dominant(kernels)

    lib.kamd_profile_reset()
    lib.kamd_profile_select(kernel_ids.get(dom, -1) if dom else -1)
    lib.kamd_profile_enable(1 if dom else 0)
    posted0 = reducer.posted
    dt = timed(dibr_step, args.steps, max(2, args.warmup))   # (untimed steps in THIS mode: its event pool is created on first use)
    reducer_posted_per_step = (reducer.posted - posted0) / (args.steps + max(2, args.warmup))
    dibr_enqueue_ms = timed.enqueue_ms
    dibr_enqueue_idle_ms = host_enqueue_idle_ms(dibr_step)
    lib.kamd_profile_enable(0)
    lib.kamd_profile_select(-1)
    dom_ms, dom_n = _lib.kernel_profile(reset=True).get(dom, (0.0, 0)) if dom else (0.0, 0)
    ms_per_step = dt / args.steps * 1e3
    mpix = world * V * H * W * args.steps / dt / 1e6

    step_stats = per_step_ms(dibr_step, max(args.steps, 20))
    if args.dump_vertex_grad:
        dibr_step()
        torch.cuda.synchronize()
        if rank == 0:
            torch.save(verts.grad.detach().cpu(), args.dump_vertex_grad)
    feature_grad = tutorial = torch_loss = None
    if not args.quick:
        # variant with gradients w.r.t. the face features as well (raster_backward adds its feature-gradient atomics)
        fg_dt = timed(dibr_step_feature_grad, args.steps, args.warmup)
        fg_stats = per_step_ms(dibr_step_feature_grad, max(args.steps, 20))
        feature_grad = {'ms_per_step': round(fg_dt / args.steps * 1e3, 4), 'per_step_ms': fg_stats,
                        'value': round(world * V * H * W * args.steps / fg_dt / 1e6, 2), 'unit': 'Mpixels/s',
                        'note': 'same step with face_features.requires_grad (gradients to vertices AND features)'}

        tl_dt = timed(dibr_step_tutorial, args.steps, args.warmup)
        tutorial = {'ms_per_step': round(tl_dt / args.steps * 1e3, 4), 'per_step_ms': per_step_ms(dibr_step_tutorial, max(args.steps, 20)),
                    'value': round(world * V * H * W * args.steps / tl_dt / 1e6, 2), 'unit': 'Mpixels/s',
                    'note': 'same step with the tutorial\'s objective: torch L1 image loss + kaolin.metrics.render.mask_iou (one fused '
                            'pass each way) instead of the two dot products'}

        tq_dt = timed(dibr_step_torch_loss, args.steps, args.warmup)
        torch_loss = {'ms_per_step': round(tq_dt / args.steps * 1e3, 4), 'per_step_ms': per_step_ms(dibr_step_torch_loss, max(args.steps, 20)),
                      'value': round(world * V * H * W * args.steps / tq_dt / 1e6, 2), 'unit': 'Mpixels/s',
                      'note': 'same step with the linear loss written in torch (two rocBLAS dots, an add, two full-size products '
                              'backward) instead of kaolin_amd.metrics.render.weighted_sum: what a drop-in user of the reference\'s '
                              'API gets without touching the loss'}

    # ---------------- the other scene: the same step on the non-convex knot (or, with --scene knot, on the
    # sphere), timed over fewer steps, with its own kernel table -- no kernel should be much slower than on the sphere
    # without an explanation in DESIGN.md
    scene_variants = None
    headline_units = work_units(scene, face_idx, p_cov)
    if not args.quick and not args.no_scene_variants:
        scene_variants = {}
        for other in ('sphere', 'knot', 'knot_shuffled'):
            if other == args.scene:
                continue
            try:
                sc = build_scene(other)
                sc['front_faces'] = front_faces(sc)
                step_o = make_step(sc, sc['feats3'])
                for _ in range(max(args.warmup // 2, 3)):
                    fi = step_o()
                n_o = max(min(args.steps, 30), 10)
                table_o, inst_o, fi, p_cov_o, over_o = kernel_table(step_o, sc, n_o)
                dt_o = timed(step_o, n_o, 2)
                st_o = per_step_ms(step_o, max(n_o, 20))
                if other == 'knot_shuffled':   # the knot with its faces in a random order: the same work, only the list order differs
                    scene_variants[other] = {
                        'faces': sc['F'], 'ms_per_step': round(dt_o / n_o * 1e3, 4), 'per_step_ms': st_o,
                        'kernels_avg_us': {k: v['avg_us'] for k, v in table_o.items()},
                        'vs_ordered_knot_step': (round(dt_o / n_o * 1e3 / scene_variants['knot']['ms_per_step'], 3)
                                                 if 'ms_per_step' in scene_variants.get('knot', {}) else None)}
                    del sc, step_o
                    continue
                wu_o = work_units(sc, fi, p_cov_o)
                wr = {k: (round(wu_o[k] / headline_units[k], 3) if headline_units[k] > 0 else None) for k in wu_o}
                scene_variants[other] = {
                    'faces': sc['F'], 'front_faces_per_view': round(sc['front_faces'], 1),
                    'work_units_per_view': wu_o, 'vs_headline_scene_work_ratio': wr,
                    # duration ratio / work ratio per kernel: ~1 = the kernel costs the same per unit of work on both scenes
                    'kernel_ratio_over_work_ratio': {k: round(v['avg_us'] / kernels[k]['avg_us'] / wr[KERNEL_WORK_UNIT[k]], 2)
                                                     for k, v in table_o.items()
                                                     if k in kernels and kernels[k]['avg_us'] > 0 and wr.get(KERNEL_WORK_UNIT.get(k))},
                    'kernel_work_unit': KERNEL_WORK_UNIT,
                    'covered_pixel_fraction': round(float((fi >= 0).float().mean()), 4),
                    'covered_tile_pixel_fraction': round(p_cov_o / (H * W), 4),
                    'ms_per_step': round(dt_o / n_o * 1e3, 4), 'per_step_ms': st_o,
                    'value': round(world * V * H * W * n_o / dt_o / 1e6, 2), 'unit': 'Mpixels/s',
                    'kernels_avg_us': {k: v['avg_us'] for k, v in table_o.items()},
                    'vs_headline_scene_kernel_ratio': {k: round(v['avg_us'] / kernels[k]['avg_us'], 2) for k, v in table_o.items()
                                                       if k in kernels and kernels[k]['avg_us'] > 0},
                    'kernels_over_stream_copy_rate': over_o}
                del sc, step_o
            except Exception as exc:                    # (must not cost the run its headline line)
                scene_variants[other] = {'error': f'{type(exc).__name__}: {str(exc)[:200]}'}
                torch.cuda.synchronize()

    # ---------------- the reference-contract operators at C4 (the eight `_C` entry points; here the four of the
    # DIB-R path with their K-buffers): the only place where 8(d)'s contract bytes -- 872 B/pixel + 296 B/face -- are
    # physically moved.  Each operator is called through kaolin_amd._C exactly as the reference's autograd Functions call
    # it (packed faces / K-buffers); durations from events around the call, inputs prepared outside.
    contract_ops = None
    if rank == 0 and not args.no_contract_ops:
        try:
            contract_ops = time_contract_operators(verts, faces, proj, rot, trans, feats3, G1, G2, H, W, max(args.steps // 2, 5))
        except Exception as exc:                        # (must not cost the run its headline line)
            contract_ops = {'error': f'{type(exc).__name__}: {str(exc)[:200]}'}
            torch.cuda.synchronize()

    # ---------------- the same step replayed as a HIP graph (N = 1: no host in it)
    graph_replay = None
    # (not under a process group, even of one rank: the collective backend's watchdog thread polls events, which HIP refuses
    # while another thread is capturing)
    if world == 1 and not D.is_distributed() and not args.quick:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    dibr_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                dibr_step()
            gdt = timed(graph.replay, args.steps, args.warmup)
            graph_replay = {'ms_per_step': round(gdt / args.steps * 1e3, 4), 'per_step_ms': per_step_ms(graph.replay, max(args.steps, 20)),
                            'note': 'the eager step above captured once with torch.cuda.graph and replayed: what the GPU needs '
                                    'when the host enqueues nothing'}
            del graph
        except Exception as exc:
            graph_replay = {'error': f'{type(exc).__name__}: {str(exc)[:200]}'}
            torch.cuda.synchronize()

    covered = float((face_idx >= 0).float().mean())
    traffic, step_traffic = None, None
    if dom and os.path.exists(tpath) and args.scene == 'sphere':   # (the counters were collected on config C4's scene)
        # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, calibrated on launches of
        # known size: tools/pmc_traffic.py, tools/parse_traffic.py, raw tables in profiles/r02*_pmc_*.txt)
        tj = json.load(open(tpath))
        traffic = (tj.get(dom) or {}).get('hbm_bytes')
        per_kernel = {k: round(v['hbm_bytes']) for k, v in tj.items() if k in kernels and v.get('hbm_bytes')}
        if per_kernel:
            st = tj.get('_step') or {}
            step_traffic = {'hbm_bytes_per_step': int(st.get('hbm_bytes') or
                                                      sum(per_kernel[k] * kernels[k]['launches_per_step'] for k in per_kernel)),
                            'of_which_other_kernels(torch dot / memset)': int(st.get('other_kernels_hbm_bytes') or 0),
                            'per_kernel_hbm_bytes_per_launch': per_kernel,
                            'kernels_without_counters': sorted(k for k in kernels if k not in per_kernel),
                            'source': tj.get('_source', 'profiles/traffic.json')}
    roofline = None
    notes = {}   # the long strings of the line, printed LAST (a reader that truncates the line's tail loses prose, not numbers)
    if dom and dom_n:
        dom_us = dom_ms / dom_n * 1e3                        # measured inside the timed region
        dom_bytes = int(round(algorithmic_bytes(dom, V, H * W, F, Fv, 3, 30, P_cov=p_cov)))
        dom_gbps = dom_bytes / (dom_us * 1e-6) / 1e9
        roofline = {'kernel': dom, 'bound': 'hbm', 'achieved': round(dom_gbps, 1), 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': round(dom_gbps / HBM_PEAK_GBS, 4),
                    'traffic': traffic if args.scene == 'sphere' else None, 'avg_launch_us': round(dom_us, 2),
                    'algorithmic_bytes_per_launch': dom_bytes,
                    # `traffic` is NOT measured in this run: it is read from the tracked file below, which a PMC pass of an earlier
                    # call wrote (its `_source` says which command, on which commit of the library)
                    'traffic_source': (f'profiles/traffic.json: {tj.get("_source", "?")}' if traffic and args.scene == 'sphere' else None)}
        if traffic and args.scene == 'sphere':
            # the same duration against the bytes the PMC counters saw the kernel move (FETCH_SIZE + WRITE_SIZE)
            roofline['frac_on_counter_bytes'] = round(traffic / (dom_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        if dom == 'raster_tile_kernel':
            # what THIS launch has to move: face_idx 8 + features 4 D + soft mask 4 bytes for every pixel,
            # the 12 bytes of weights only in the tiles that hold a covered pixel (background tiles leave them unwritten), the
            # front faces' 88 bytes
            launch_bytes = int(round(V * (H * W * (8 + 4 * 3 + 4) + p_cov * 12 + Fv * 88)))
            roofline['launch_bytes'] = launch_bytes
            roofline['frac_on_launch_bytes'] = round(launch_bytes / (dom_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            notes['roofline.bytes_note'] = ('algorithmic bytes (achieved / frac) = SURVEY 8(d) K1: 32 B/pixel (face_idx i64 + 3 weights + 3 features) + 88 B '
                                      'per front face; the fused launch also writes 4 B/pixel of soft mask and leaves the 12 B/pixel of '
                                      'weights unwritten in tiles without a covered pixel (internal to the autograd node): launch_bytes / '
                                      'frac_on_launch_bytes count exactly what it writes and reads; frac_on_counter_bytes = the PMC bytes')
    # whole-step figure against the contract bytes of SURVEY.md 8(d): 872 B/pixel + 296 B/face (D=3, K=30, fp32)
    contract = V * (H * W * 872 + F * 296)
    lean = V * (H * W * 48 + F * 136)
    step_roofline = {'contract_bytes_per_step': contract, 'contract_GBps': round(contract / (ms_per_step * 1e-3) / 1e9, 1),
                     'contract_frac_of_8TBps': round(contract / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     'lean_bytes_per_step': lean}
    if step_traffic:
        step_roofline['measured_hbm_GBps'] = round(step_traffic['hbm_bytes_per_step'] / (ms_per_step * 1e-3) / 1e9, 1)

    # ---------------- chamfer 100k x 100k fwd+bwd (config C3: one batch item per rank + shared offset)
    chamfer = None
    if not args.no_chamfer:
        n = args.chamfer_points
        gen = torch.Generator().manual_seed(rank)
        base = torch.rand((1, n, 3), generator=gen).to(dev)
        p2 = torch.rand((1, n, 3), generator=gen).to(dev).requires_grad_()
        offset = torch.zeros(3, device=dev, requires_grad=True)

        creducer = D.SharedGradientReducer([offset])

        def chamfer_step():
            offset.grad = None
            p2.grad = None
            kal.metrics.pointcloud.chamfer_distance(base + offset, p2).sum().backward()
            creducer.wait()

        cdt = timed(chamfer_step, args.steps, args.warmup)
        chamfer_enqueue_ms = timed.enqueue_ms
        chamfer_enqueue_idle_ms = host_enqueue_idle_ms(chamfer_step)
        lib.kamd_profile_reset()
        lib.kamd_profile_enable(1)                 # per-kernel durations: a separate, instrumented pass
        timed(chamfer_step, args.steps, 0)
        lib.kamd_profile_enable(0)
        cprof = _lib.kernel_profile(reset=True)
        pairs = 2.0 * world * n * n * args.steps
        chamfer = {'metric': 'Mpoint-pairs/s chamfer fwd+bwd (effective pairs = 2*N*M per item; the exact grid search '
                             'evaluates far fewer and returns the brute-force-identical result)',
                   'value': round(pairs / cdt / 1e6, 1), 'ms_per_step': round(cdt / args.steps * 1e3, 4), 'points': n,
                   'host_enqueue_ms_per_step': round(chamfer_enqueue_ms, 4),
                   'host_enqueue_gpu_idle_ms_per_step': round(chamfer_enqueue_idle_ms, 4),
                   'kernels_avg_us': {k: round(v[0] / v[1] * 1e3, 2) for k, v in cprof.items()},
                   'hbm_frac_of_8TBps': round(192.0 * n / (cdt / args.steps) / 1e9 / HBM_PEAK_GBS, 6)}
        # roofline of the search launch (the largest kernel of the chamfer step): SURVEY 8(d)'s forward bytes -- 36 B per point and
        # direction -- over its event-bracketed duration, and the PMC bytes of the same launch (profiles/traffic.json, section
        # _chamfer_step; collected on the C3 size only)
        q_us = chamfer['kernels_avg_us'].get('sdg_query')
        if q_us:
            q_bytes = 2 * 36 * n
            tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
            sec = (tj.get('_chamfer_step') or {}) if n == 100000 else {}
            q_cnt = (sec.get('kernels') or {}).get('sdg_query')
            q_traffic = (q_cnt['fetch_bytes_per_call'] + q_cnt['write_bytes_per_call']) / max(q_cnt['launches_per_call'], 1) if q_cnt else None
            chamfer['roofline'] = {'kernel': 'sdg_query', 'bound': 'hbm', 'achieved': round(q_bytes / (q_us * 1e-6) / 1e9, 1), 'peak': HBM_PEAK_GBS,
                                   'unit': 'GB/s', 'frac': round(q_bytes / (q_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                                   'avg_launch_us': q_us, 'algorithmic_bytes_per_launch': q_bytes,
                                   'traffic': None if q_traffic is None else int(q_traffic),
                                   'frac_on_counter_bytes': None if q_traffic is None else round(q_traffic / (q_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                                   'step_counter_bytes': int(sec['hbm_bytes_per_call']) if sec.get('hbm_bytes_per_call') else None,
                                   'step_frac_on_counter_bytes': (round(sec['hbm_bytes_per_call'] / (cdt / args.steps) / 1e9 / HBM_PEAK_GBS, 5)
                                                                  if sec.get('hbm_bytes_per_call') else None),
                                   'note': 'an exact nearest-neighbour search is a latency chain over ~60 candidate targets per query, not a '
                                           'stream: the HBM fraction says how far from memory-bound it is (SURVEY 8(d): brute force is VALU-bound '
                                           'at 8 300 FLOP/B)'}
        # The step above wraps the operator in torch glue for the shared parameter (base + offset, .sum(), their backward
        # nodes and the reduction to 3 floats): ~10 small host-bound torch calls.  Two more readings of the same work:
        # (a) the operator alone -- chamfer_distance forward + backward to both clouds from a given upstream gradient,
        # no collective; (b) the step above captured once with torch.cuda.graph and replayed (N = 1 only: no host in it).
        p1_leaf = base.clone().requires_grad_()
        upstream = torch.ones(1, device=dev)

        def chamfer_operator():
            p1_leaf.grad = None
            p2.grad = None
            kal.metrics.pointcloud.chamfer_distance(p1_leaf, p2).backward(upstream)

        odt = timed(chamfer_operator, args.steps, args.warmup)
        chamfer['operator_only'] = {'ms_per_step': round(odt / args.steps * 1e3, 4),
                                    'host_enqueue_ms_per_step': round(timed.enqueue_ms, 4),
                                    'value': round(pairs / odt / 1e6, 1),
                                    'note': 'chamfer_distance fwd + bwd (gradients to both clouds) from a given upstream '
                                            'gradient; no shared parameter, no collective'}
        if world == 1 and not D.is_distributed():
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        chamfer_step()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    chamfer_step()
                gdt = timed(graph.replay, args.steps, args.warmup)
                chamfer['graph_replay_ms_per_step'] = round(gdt / args.steps * 1e3, 4)
                del graph
            except Exception as exc:                    # (a capture failure must not cost the run its headline line)
                chamfer['graph_replay_ms_per_step'] = None
                chamfer['graph_replay_error'] = str(exc)[:200]
                torch.cuda.synchronize()
        # SURVEY 8(d) C3's single-GPU reference run: the same 8 items (item r: torch.manual_seed(r), two rand(1, n, 3) draws) as ONE
        # B = 8 call on this GPU, the shared offset's gradient summed by autograd instead of by the all-reduce.  One item is
        # 2.4 MB of input and bounded by launch latency; eight in one launch show what the search does with a full machine.
        if world == 1:
            b8 = 8
            items = []
            for r in range(b8):
                g8 = torch.Generator().manual_seed(r)
                items.append((torch.rand((1, n, 3), generator=g8), torch.rand((1, n, 3), generator=g8)))
            base8 = torch.cat([it[0] for it in items], 0).to(dev)
            p28 = torch.cat([it[1] for it in items], 0).to(dev).requires_grad_()
            offset8 = torch.zeros(3, device=dev, requires_grad=True)

            def chamfer_batch8_step():
                offset8.grad = None
                p28.grad = None
                kal.metrics.pointcloud.chamfer_distance(base8 + offset8, p28).sum().backward()

            b8dt = timed(chamfer_batch8_step, args.steps, args.warmup)
            lib.kamd_profile_reset()
            lib.kamd_profile_enable(1)
            timed(chamfer_batch8_step, args.steps, 0)
            lib.kamd_profile_enable(0)
            b8prof = _lib.kernel_profile(reset=True)
            b8_bytes = 192.0 * b8 * n
            chamfer['batch8'] = {'items': b8, 'ms_per_step': round(b8dt / args.steps * 1e3, 4),
                                 'value': round(2.0 * b8 * n * n * args.steps / b8dt / 1e6, 1),
                                 'ms_per_item': round(b8dt / args.steps * 1e3 / b8, 4),
                                 'kernels_avg_us': {k: round(v[0] / v[1] * 1e3, 2) for k, v in b8prof.items()},
                                 'roofline': {'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBS, 'algorithmic_bytes_per_step': int(b8_bytes),
                                              'achieved': round(b8_bytes / (b8dt / args.steps) / 1e9, 1),
                                              'frac': round(b8_bytes / (b8dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5)},
                                 'note': 'the 8 items of C3 as one B = 8 chamfer_distance fwd + bwd on one GPU (SURVEY 8(d): the run the sharded '
                                         'result must match; tests/test_full_size_parity.py::test_c3_batch8_equals_sharded_items)'}
            del base8, p28, offset8, items
        # the all-pairs kernels for comparison (VALU-bound: 6.7 lane-ops per pair, sided_distance.hip header)
        os.environ['KAMD_SIDED_DISTANCE'] = 'brute'
        lib.kamd_profile_reset()
        lib.kamd_profile_enable(1)
        bdt = timed(chamfer_step, max(args.steps // 4, 2), 2)
        lib.kamd_profile_enable(0)
        bprof = _lib.kernel_profile(reset=True)
        del os.environ['KAMD_SIDED_DISTANCE']
        main_ms, main_n = bprof.get('sd_main_f32', (0.0, 1))
        chamfer['brute_force'] = {'value': round(2.0 * world * n * n * max(args.steps // 4, 2) / bdt / 1e6, 1),
                                  'sd_main_avg_us': round(main_ms / max(main_n, 1) * 1e3, 2),
                                  'valu_Tlaneops_per_s': round(6.7 * n * n / (main_ms / max(main_n, 1) * 1e-3) / 1e12, 2) if main_ms else None}

    # ---------------- config C5 extras (rank 0 only, replicas-only ops): voxelgrid 256^3 + point_to_mesh 1M x 50k
    c5 = None
    if rank == 0 and not args.no_c5:
        def per_call_ms(fn, n=5):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        v1 = verts.detach().unsqueeze(0)
        vox_ms = per_call_ms(lambda: kal.ops.conversions.trianglemeshes_to_voxelgrids(v1, faces, 256), 10)
        fv = v1[0][faces].unsqueeze(0).contiguous()
        q = (torch.rand((1, 1000000, 3), generator=torch.Generator().manual_seed(0)) * 1.2 - 0.6).to(dev)
        p2m_ms = per_call_ms(lambda: kal.metrics.trianglemesh.point_to_mesh_distance(q, fv), 3)
        lib.kamd_profile_reset()
        lib.kamd_profile_enable(1)                 # kernel durations of both: a separate, instrumented pass
        for _ in range(5):
            kal.ops.conversions.trianglemeshes_to_voxelgrids(v1, faces, 256)
        kal.metrics.trianglemesh.point_to_mesh_distance(q, fv)
        torch.cuda.synchronize()
        lib.kamd_profile_enable(0)
        kprof = {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()}
        vox_kernels = {k: v for k, v in kprof.items() if k.startswith('vox')}
        c5 = {'voxelgrid_256_us': round(vox_ms * 1e3, 1), 'voxelgrid_write_GBps': round(256 ** 3 * 4 / (vox_ms * 1e-3) / 1e9, 1),
              'voxelgrid_kernels_avg_us': vox_kernels,
              'voxelgrid_kernels_sum_frac_of_write_bound': round(256 ** 3 * 4 / 8e12 * 1e6 / max(sum(vox_kernels.values()), 1e-3), 3),
              'roofline': None,
              'point_to_mesh_1Mx50k_ms': round(p2m_ms, 3),
              'point_to_mesh_kernels_avg_us': {k: v for k, v in kprof.items() if k.startswith('td_')},
              'point_to_mesh_Gpairs_per_s': round(1e6 * F / (p2m_ms * 1e-3) / 1e9, 1),
              # the all-pairs kernel issues 11 VALU lane-ops per (point, face) sphere test at 58 T lane-ops/s measured
              # (profiles/r01_ubench_valu.txt): > 1 means the exact search evaluated that much less than all pairs
              'point_to_mesh_allpairs_equiv_valu_frac': round(11.0 * 1e6 * F / (p2m_ms * 1e-3) / 58e12, 3)}
        # What the exact search really evaluates, against the fp32 vector peak ("1.87x the all-pairs equivalent" is a
        # speed-up, not a roofline fraction).  The sweep kernel's own work counters (kamd_triangle_distance_work_counters: one extra,
        # counting call): closest-point evaluations (the reference's per-pair arithmetic, unbatched_triangle_distance_cuda.cu:237-317:
        # ~80 FLOP with its three divisions), face steps (a wavefront tests one face's bounding sphere / plane slab against its 64
        # queries: ~10 FLOP per lane) and (query, tile) sphere tests (~10 FLOP); over td_main's duration and 157.3 TFLOP/s.
        try:
            cnt = (ctypes.c_ulonglong * 8)()
            lib.kamd_triangle_distance_work_counters(1, None)
            kal.metrics.trianglemesh.point_to_mesh_distance(q, fv)
            torch.cuda.synchronize()
            lib.kamd_triangle_distance_work_counters(0, cnt)
            evals, face_steps, tile_tests = int(cnt[4]), int(cnt[3]), int(cnt[2])
            flop = 80.0 * evals + 10.0 * 64.0 * face_steps + 10.0 * tile_tests
            main_us = kprof.get('td_main_kernel') or kprof.get('td_main') or 0.0
            c5['point_to_mesh_work'] = {'closest_point_evaluations': evals, 'face_steps_per_wavefront': face_steps, 'query_tile_tests': tile_tests,
                                        'hard_queries': int(cnt[7]), 'flop_model': '80 per evaluation + 10 per lane and face step + 10 per (query, tile) test',
                                        'flop': flop, 'all_pairs': int(1e6 * F), 'evaluated_fraction_of_all_pairs': round(evals / (1e6 * F), 6)}
            c5['point_to_mesh_valu_frac'] = round(flop / (main_us * 1e-6) / 157.3e12, 5) if main_us else None
        except Exception as exc:                        # (a missing entry point must not cost the run its line)
            c5['point_to_mesh_valu_frac'] = None
            c5['point_to_mesh_work'] = {'error': f'{type(exc).__name__}: {str(exc)[:160]}'}
        # roofline of the voxelizer (genuinely HBM-write-bound: SURVEY 8(d)): the dense grid's bytes over the two launches' summed
        # durations, and the PMC bytes of one call (profiles/traffic.json, section _voxelgrid_256)
        vox_us = sum(vox_kernels.values())
        if vox_us > 0:
            tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
            sec = tj.get('_voxelgrid_256') or {}
            vb = 256 ** 3 * 4 + verts.shape[0] * 12 + F * 24
            c5['roofline'] = {'kernel': 'vox_clear_extent_kernel + vox_mark_kernel', 'bound': 'hbm', 'achieved': round(vb / (vox_us * 1e-6) / 1e9, 1),
                              'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(vb / (vox_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                              'avg_launch_us': round(vox_us, 1), 'algorithmic_bytes_per_launch': vb,
                              'traffic': int(sec['hbm_bytes_per_call']) if sec.get('hbm_bytes_per_call') else None,
                              'frac_on_counter_bytes': (round(sec['hbm_bytes_per_call'] / (vox_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                                                        if sec.get('hbm_bytes_per_call') else None)}
        # deftet_sparse_render fwd+bwd, view 0 of the same mesh, knum 30, free pixel coordinates
        with torch.no_grad():
            d_cam, d_img, _ = kal.render.mesh.prepare_vertices(
                verts.detach().unsqueeze(0), faces, proj, camera_rot=rot[:1], camera_trans=trans[:1])
        d_z, d_img, d_feat = d_cam[..., 2].contiguous(), d_img.contiguous(), feats3[:1].contiguous()
        for n_pix in (4096, 1 << 20):
            gen = torch.Generator().manual_seed(1)
            pix = (torch.rand((1, n_pix, 2), generator=gen) * 2. - 1.).to(dev)
            rng = torch.tensor([[[-10., 0.]]], device=dev).repeat(1, n_pix, 1)
            g_out = torch.rand((1, n_pix, 30, 3), generator=gen).to(dev)
            a_img, a_feat = d_img.clone().requires_grad_(), d_feat.clone().requires_grad_()

            def deftet_step():
                a_img.grad = None
                a_feat.grad = None
                out, _ = kal.render.mesh.deftet_sparse_render(pix, rng, d_z, a_img, a_feat, 30)
                out.backward(g_out)
            deftet_step()          # first call: module load / allocator growth stay out of the event timings
            lib.kamd_profile_reset()
            lib.kamd_profile_enable(1)
            ms = per_call_ms(deftet_step, 5)
            lib.kamd_profile_enable(0)
            prof = {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items() if 'deftet' in k}
            c5[f'deftet_{n_pix}px_x_{F}f_fwd_bwd_ms'] = round(ms, 3)
            c5[f'deftet_{n_pix}px_kernels_avg_us'] = prof

    # ---------------- CPU baseline: the oracle (OpenMP) on a bounded sample, rank 0 at N = 1 only
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        import oracle
        oracle.build()
        with torch.no_grad():
            fv_cam, fv_img, normals = kal.render.mesh.prepare_vertices(
                verts.detach().unsqueeze(0), faces, proj, camera_rot=rot[:1], camera_trans=trans[:1])
        fz, fimg, nz = fv_cam[..., 2].cpu(), fv_img.cpu(), normals[..., 2].cpu()
        feat = feats3[:1].cpu()

        # The forward AND the two backward passes run OpenMP over pixels (round 6; VERDICT r05: a line that says N cores must not contain
        # a serial leg).  How many threads: the OpenMP runtime's default, the process's affinity mask and os.cpu_count() can all differ
        # (round-6 boxes: 128 / 256 / 256, and 256 spinning threads under the container's CPU quota ran the forward THIRTY times
        # slower than 128) -- so every distinct candidate is probed on a small image and the fastest is used; `cores` = that count.
        cand = {oracle.num_threads(True), os.cpu_count() or 1}
        if hasattr(os, 'sched_getaffinity'):
            cand.add(len(os.sched_getaffinity(0)))
        probes = {}
        if len(cand) > 1:
            for nthr in sorted(cand):
                oracle.set_num_threads(nthr)
                t0 = time.perf_counter()
                oracle.dibr_rasterization(64, 64, fz, fimg, feat, nz, omp=True)
                probes[nthr] = round(time.perf_counter() - t0, 3)
            oracle.set_num_threads(min(probes, key=probes.get))
        cpu_legs = {'forward_s': 0.0, 'backward_s': 0.0}

        def cpu_pass(res):
            t0 = time.perf_counter()
            ref = oracle.dibr_rasterization(res, res, fz, fimg, feat, nz, omp=True)
            t1 = time.perf_counter()
            oracle.rasterize_backward(torch.ones_like(ref['features']), ref['face_idx'], ref['weights'], fimg, feat, 1e-8, omp=True)
            oracle.dibr_soft_mask_backward(torch.ones_like(ref['soft_mask']), ref['soft_mask'], ref['face_idx'],
                                           ref['close_face_prob'], ref['close_face_idx'], ref['close_face_dist_type'],
                                           ref['scaled_vertices'], 7000, 1000., omp=True)
            t2 = time.perf_counter()
            cpu_legs['forward_s'] += t1 - t0
            cpu_legs['backward_s'] += t2 - t1
            return t2 - t0

        probe = cpu_pass(128)                                  # sizes the sample for ~15 s of CPU work
        sres = int(min(1024, max(128, 128 * math.sqrt(15.0 / max(probe, 1e-3)))) // 32 * 32)
        cpu_legs['forward_s'] = cpu_legs['backward_s'] = 0.0
        cdt = cpu_pass(sres)
        reps = 1
        if cdt < 10.0:                                         # many host cores: repeat the view until ~12 s are spent
            more = min(int(math.ceil(12.0 / cdt)) - 1, 15)
            for _ in range(more):
                cdt += cpu_pass(sres)
            reps += more
        # `cores` = the threads a reading actually used, everywhere in this object (the OpenMP port: every host core; the torch oracles
        # of other_paths: at most 16, their dense temporaries scale badly beyond); `host_cores` = what the box has
        cpu = {'value': round(reps * sres * sres / cdt / 1e6, 4), 'unit': 'Mpixels/s', 'cores': oracle.num_threads(True),
               'kind': 'port', 'host_cores': os.cpu_count(),
               'forward_s': round(cpu_legs['forward_s'], 3), 'backward_s': round(cpu_legs['backward_s'], 3),
               'thread_count_probes_s': probes or None,
               'sample': f'{reps} pass(es) over 1 view of the same {F}-triangle mesh at {sres}x{sres} (the brute-force reference algorithm costs '
                         f'O(faces) per pixel at any resolution), oracle forward + both backward passes, all three OpenMP over pixels on '
                         f'{oracle.num_threads(True)} threads (the backward passes add their terms with atomic double adds), {cdt:.1f} s '
                         f'= forward {cpu_legs["forward_s"]:.1f} + backward {cpu_legs["backward_s"]:.2f}'}
        if not args.no_cpu_extras:
            # BASELINE.md section 3's other CPU readings (the torch formulations the reference's tests use as oracles, and the
            # restated chamfer kernel), each on a bounded sample.  In a child process without a GPU and with a hard limit:
            # whatever the host does with them, the run keeps its headline line.
            import subprocess
            try:
                res = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-extras-only', '--chamfer-points',
                                      str(args.chamfer_points), '--sphere-frequency', str(args.sphere_frequency)],
                                     env=dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES=''),
                                     capture_output=True, text=True, timeout=75)
                cpu['other_paths'] = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1])
            except Exception as exc:
                cpu['other_paths'] = {'error': f'{type(exc).__name__}: {str(exc)[:160]}'}

    if rank == 0:
        # The honest spellings of the headline and the secondary paths' one-number summaries, where a reader of the driver's record
        # sees them (the driver keeps `roofline`, `cpu_baseline` and `config` whole and cuts every other key to its
        # name): nested in `roofline.same_run`, and once more as top-level `also_*` scalars straight after the two objects.
        headline_scalars = {
            'torch_loss_variant_value': torch_loss['value'] if torch_loss else None,            # the loss in plain torch: what a user of the reference's API gets
            'feature_grad_variant_value': feature_grad['value'] if feature_grad else None,      # + gradients w.r.t. the face features
            'tutorial_loss_variant_value': tutorial['value'] if tutorial else None,             # the DIB-R tutorial's objective
            'median_ms_per_step': step_stats['median'],
            'host_enqueue_gpu_idle_ms_per_step': round(dibr_enqueue_idle_ms, 4),
            'graph_replay_ms_per_step': (graph_replay or {}).get('ms_per_step'),
            'knot_scene_ms_per_step': ((scene_variants or {}).get('knot') or {}).get('ms_per_step'),
            'knot_shuffled_scene_ms_per_step': ((scene_variants or {}).get('knot_shuffled') or {}).get('ms_per_step'),
            'chamfer_step_ms': (chamfer or {}).get('ms_per_step'),
            'chamfer_operator_ms': ((chamfer or {}).get('operator_only') or {}).get('ms_per_step'),
            'chamfer_batch8_ms': ((chamfer or {}).get('batch8') or {}).get('ms_per_step'),
            'chamfer_batch8_hbm_frac': (((chamfer or {}).get('batch8') or {}).get('roofline') or {}).get('frac'),
            'c5_voxelgrid_256_us': (c5 or {}).get('voxelgrid_256_us'),
            'c5_point_to_mesh_ms': (c5 or {}).get('point_to_mesh_1Mx50k_ms'),
            'c5_point_to_mesh_valu_frac': (c5 or {}).get('point_to_mesh_valu_frac'),
        }
        out = {
            'metric': 'Mpixels/s DIB-R fwd+bwd @1024^2', 'value': round(mpix, 2), 'unit': 'Mpixels/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            # beside the K-step mean of the contract: the per-step MEDIAN of one more pass (SURVEY 8(d) asks for the median; a
            # 20-step driver run is otherwise at the mercy of one outlier) and the same step with the loss written in plain torch
            # (what a drop-in user of the reference's API gets: kaolin has no fused weighted_sum)
            'median_ms_per_step': step_stats['median'],
            'value_at_median_ms_per_step': round(world * V * H * W / (step_stats['median'] * 1e-3) / 1e6, 2),
            'torch_loss_value': torch_loss['value'] if torch_loss else None,
            'torch_loss_median_ms_per_step': torch_loss['per_step_ms']['median'] if torch_loss else None,
            'feature_grad_value': feature_grad['value'] if feature_grad else None,
            'tutorial_loss_value': tutorial['value'] if tutorial else None,
            'host_enqueue_ms_per_step': round(dibr_enqueue_ms, 4),
            'host_enqueue_gpu_idle_ms_per_step': round(dibr_enqueue_idle_ms, 4),   # (the host's own cost: see host_enqueue_idle_ms)
            'graph_replay_ms_per_step': (graph_replay or {}).get('ms_per_step'),
            'roofline': None if roofline is None else dict(roofline, same_run=headline_scalars),
            'cpu_baseline': None if cpu is None else {k: v for k, v in cpu.items() if k not in ('other_paths', 'sample')} | {'sample': cpu['sample']},
            **{'also_' + k: v for k, v in headline_scalars.items()},
            'config': {'workload': f'{"C4" if args.scene == "sphere" else "C4 shape, scene " + args.scene}: dibr_rasterization fwd+bwd, {V} views/GPU at {H}x{W} of a {F}-triangle '
                                   f'{"geodesic sphere" if args.scene == "sphere" else "non-convex knot scene (kaolin_amd.utils.testing.knot_mesh)" + (", faces in random order" if args.scene == "knot_shuffled" else "")} '
                                   f'(shared vertices), D=3 static face features (uv + mask channel; gradient w.r.t. the vertices only), '
                                   f'knum=30, sigmainv=7000, boxlen=0.02, loss = sum(features*G1) + sum(soft_mask*G2) (fused weighted_sum), '
                                   f'prepare_vertices + vertex-gradient all-reduce (posted from the autograd hook) inside the step',
                       'views_per_gpu': V, 'global_views': V * world, 'height': H, 'width': W, 'faces': F,
                       'scene': args.scene, 'front_faces_per_view': round(Fv, 1),
                       'covered_pixel_fraction': round(covered, 4), 'covered_tile_pixel_fraction': round(p_cov / (H * W), 4),
                       'parallelism': f'views sharded {world}-way', 'look_at': list(args.look_at)},
            'per_step_ms': step_stats, 'kernels': kernels, 'step_roofline': step_roofline,
            'distributed': {'initialized': D.is_distributed(),
                            'backend': torch.distributed.get_backend() if D.is_distributed() else None,
                            'collectives_posted_per_step': round(reducer_posted_per_step, 3),
                            'visible_gpus': torch.cuda.device_count(),
                            'ranks_per_gpu': max(1, -(-world // max(torch.cuda.device_count(), 1)))},
            'chamfer': chamfer, 'c5': c5,
            'feature_grad_variant': feature_grad, 'tutorial_loss_variant': tutorial, 'torch_loss_variant': torch_loss,
            'work_units_per_view': headline_units, 'scene_variants': scene_variants,
            'step_traffic': step_traffic,
            'kernels_over_stream_copy_rate': over_peak,    # (algorithmic_GBps above what a streaming copy reaches = a byte model that is wrong)
            'instrumented_ms_per_step': round(inst_ms_per_step, 4),
            'graph_replay': graph_replay, 'contract_operators': contract_ops,
            'cpu_baseline_other_paths': (cpu or {}).get('other_paths'),
            'notes': dict(notes, kernels_note=f'per-kernel table: separate pass of {args.steps} steps with every launch timed by HIP events '
                                              f'(hipExtLaunchKernelGGL start / stop events = the dispatch\'s own timestamps for the DIB-R kernels, two '
                                              f'records around multi-launch operators; {inst_ms_per_step:.4f} ms/step: the events and the single-stream '
                                              f'order they need cost the difference); the timed region times the roofline kernel only'),
        }
        print(json.dumps(out))
    if D.is_distributed():
        D.barrier()     # rank 0 still runs the C5 extras and prints the line: the group goes down only when every rank is done
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
