"""N > 1 path on CPU: two gloo processes.  Config C3's structure: every rank owns some batch
items of a chamfer problem, a parameter `shared_offset` is replicated, per-item gradients stay local and ONE
all-reduce carries the shared gradient; the sharded result must equal the single-process full-batch result.
The per-item compute here is the CPU oracle (tests may use it); on the GPU box the same harness calls the HIP ops."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _chamfer_grad_wrt_offset(base, p2, offset):
    """d/d(offset) of sum_b chamfer(base + offset, p2) via the oracle's sided-distance backward."""
    import oracle
    p1 = base + offset
    B, N, M = p1.shape[0], p1.shape[1], p2.shape[1]
    d12, i12 = oracle.sided_distance_forward(p1, p2)
    d21, i21 = oracle.sided_distance_forward(p2, p1)
    g1a, _ = oracle.sided_distance_backward(torch.full((B, N), 1.0 / N, dtype=p1.dtype), p1, p2, i12)
    _, g1b = oracle.sided_distance_backward(torch.full((B, M), 1.0 / M, dtype=p1.dtype), p2, p1, i21)
    loss = d12.mean(-1).sum() + d21.mean(-1).sum()
    return (g1a + g1b).sum(dim=(0, 1)), loss


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from kaolin_amd import distributed as D
    assert D.init_from_env('gloo')
    assert D.rank() == rank and D.world_size() == world
    torch.manual_seed(0)
    B = 5  # ragged: 3 + 2 items
    base, p2 = torch.rand(B, 300, 3, dtype=torch.double), torch.rand(B, 200, 3, dtype=torch.double)
    offset = torch.zeros(3, dtype=torch.double, requires_grad=True)
    b, e = D.shard_range(B)
    assert (b, e) == ((0, 3) if rank == 0 else (3, 5))
    g, loss = _chamfer_grad_wrt_offset(D.shard(base), D.shard(p2), offset.detach())
    offset.grad = g.clone()
    unused = torch.zeros(2, requires_grad=True, dtype=torch.double)  # a parameter with no grad on this rank
    D.all_reduce_gradients([offset, unused])
    gathered = D.all_gather_batch(torch.full((2, 1), float(rank)))
    # a single shared tensor takes the in-place path (no bucket): same sum, and the average
    single = torch.zeros(3, dtype=torch.double, requires_grad=True)
    single.grad = g.clone()
    D.all_reduce_gradients([single])
    mean = torch.zeros(3, dtype=torch.double, requires_grad=True)
    mean.grad = g.clone()
    D.all_reduce_gradients([mean], average=True)
    D.barrier()
    if rank == 0:
        torch.save({'grad': offset.grad, 'unused': unused.grad, 'gathered': gathered, 'single': single.grad,
                    'mean': mean.grad}, out)
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_sharded_chamfer_matches_full_batch(tmp_path):
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    torch.manual_seed(0)
    base, p2 = torch.rand(5, 300, 3, dtype=torch.double), torch.rand(5, 200, 3, dtype=torch.double)
    g_full, _ = _chamfer_grad_wrt_offset(base, p2, torch.zeros(3, dtype=torch.double))
    assert torch.allclose(res['grad'], g_full, rtol=1e-12, atol=1e-14)
    assert torch.equal(res['unused'], torch.zeros(2, dtype=torch.double))
    assert torch.equal(res['single'], res['grad']) and torch.allclose(res['mean'], g_full / 2, rtol=1e-12, atol=1e-14)
    assert torch.equal(res['gathered'], torch.tensor([[0.], [0.], [1.], [1.]]))


def test_shard_range_covers_everything():
    from kaolin_amd import distributed as D
    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_a_noop():
    from kaolin_amd import distributed as D
    assert not D.is_distributed() and D.world_size() == 1 and D.rank() == 0
    p = torch.ones(3, requires_grad=True)
    p.grad = torch.full((3,), 2.0)
    D.all_reduce_gradients([p])
    assert torch.equal(p.grad, torch.full((3,), 2.0))
    assert D.shard(torch.arange(10)).tolist() == list(range(10))


# ---- config C4's structure: the views of one mesh split over the ranks, shared vertices, gradient reduced from a hook ----
def _view_loss(verts, cams):
    """A differentiable stand-in for `render the mesh from these views`: per view, project and weigh the vertices."""
    rel = verts.unsqueeze(0) - cams.unsqueeze(1)                    # (views, V, 3)
    return (rel[..., :2] / (1.0 + rel[..., 2:].abs())).pow(2).sum() + 0.1 * (rel.norm(dim=-1)).sum()


def _views_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from kaolin_amd import distributed as D
    assert D.init_from_env('gloo')
    torch.manual_seed(1)
    verts = torch.rand(50, 3, dtype=torch.double, requires_grad=True)
    tex = torch.rand(4, dtype=torch.double, requires_grad=True)
    cams, targets = torch.rand(7, 3, dtype=torch.double) + 2.0, torch.rand(7, 4, dtype=torch.double)
    my_cams, my_targets = D.shard_views(cams, targets)
    assert my_cams.shape[0] == (4 if rank == 0 else 3) and torch.equal(my_cams, cams[:4] if rank == 0 else cams[4:])
    reducer = D.SharedGradientReducer([verts, tex])
    loss = _view_loss(verts, my_cams) + ((tex - my_targets) ** 2).sum()
    loss.backward()
    assert reducer.posted == 2          # one collective per shared parameter, posted from inside backward()
    reducer.wait()
    reducer.remove()
    per_param = {'verts': verts.grad.clone(), 'tex': tex.grad.clone()}
    # the same with ONE collective for both parameters ("a single all-reduce", also with a trained texture)
    verts.grad = tex.grad = None
    bucket = D.SharedGradientReducer([verts, tex], single_bucket=True)
    (_view_loss(verts, my_cams) + ((tex - my_targets) ** 2).sum()).backward()
    assert bucket.posted == 1           # posted from the hook of the last gradient to arrive
    bucket.wait()
    assert torch.equal(verts.grad, per_param['verts']) and torch.equal(tex.grad, per_param['tex'])
    # ... and when one of them receives no gradient in a pass: wait() posts the bucket with zeros in its place
    verts.grad = tex.grad = None
    _view_loss(verts, my_cams).backward()
    assert bucket.posted == 1
    bucket.wait()
    assert bucket.posted == 2 and torch.equal(verts.grad, per_param['verts']) and float(tex.grad.abs().sum()) == 0.0
    bucket.remove()
    verts.grad, tex.grad = per_param['verts'], per_param['tex']
    if rank == 0:
        torch.save({'verts': verts.grad, 'tex': tex.grad}, out)
    D.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_view_sharding_with_hooked_all_reduce(tmp_path):
    out = str(tmp_path / 'views.pt')
    mp.spawn(_views_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    torch.manual_seed(1)
    verts = torch.rand(50, 3, dtype=torch.double, requires_grad=True)
    tex = torch.rand(4, dtype=torch.double, requires_grad=True)
    cams, targets = torch.rand(7, 3, dtype=torch.double) + 2.0, torch.rand(7, 4, dtype=torch.double)
    (_view_loss(verts, cams) + ((tex - targets) ** 2).sum()).backward()
    assert torch.allclose(res['verts'], verts.grad, rtol=1e-12, atol=1e-14)
    assert torch.allclose(res['tex'], tex.grad, rtol=1e-12, atol=1e-14)


# ---- the same on the GPU box: two processes share GPU 0, each renders half of the views with the HIP operators ---------
def _dibr_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import kaolin_amd as kal
    from kaolin_amd import distributed as D
    assert D.init_from_env('gloo')
    res = _dibr_views(kal, D)
    if rank == 0:
        torch.save({k: v.cpu() for k, v in res.items()}, out)
    D.barrier()
    torch.distributed.destroy_process_group()


def _dibr_views(kal, D):
    import math
    from kaolin_amd.utils import testing as T
    dev = torch.device('cuda', 0)
    V, H, W = 6, 96, 80
    v, f = T.geodesic_sphere(8)
    verts = v.float().to(dev).requires_grad_()
    faces = f.to(dev)
    cams = T.fibonacci_cameras(V, 2.5).to(dev)
    g = torch.Generator().manual_seed(3)
    G1, G2 = torch.rand((V, H, W, 3), generator=g).to(dev), torch.rand((V, H, W), generator=g).to(dev)
    uv = torch.rand((1, faces.shape[0], 3, 3), generator=g).to(dev)
    my_cams, my_G1, my_G2 = D.shard_views(cams, G1, G2) if D.is_distributed() else (cams, G1, G2)
    n = my_cams.shape[0]
    rot, trans = kal.render.camera.generate_rotate_translate_matrices(
        my_cams, torch.zeros_like(my_cams), torch.tensor([[0., 1., 0.]], device=dev).repeat(n, 1))
    proj = kal.render.camera.generate_perspective_projection(math.pi / 4).to(dev)
    reducer = D.SharedGradientReducer([verts])
    fv_cam, fv_img, normals = kal.render.mesh.prepare_vertices(verts.unsqueeze(0).expand(n, -1, -1), faces, proj,
                                                              camera_rot=rot, camera_trans=trans)
    feat, soft, face_idx = kal.render.mesh.dibr_rasterization(H, W, fv_cam[..., 2], fv_img, uv.expand(n, -1, -1, -1).contiguous(),
                                                              normals[..., 2])
    ((feat * my_G1).sum() + (soft * my_G2).sum()).backward()
    reducer.wait()
    reducer.remove()
    return {'grad': verts.grad, 'face_idx': face_idx, 'soft': soft.detach()}


@pytest.mark.gpu
def test_two_process_dibr_view_sharding_matches_single_process(tmp_path):
    """Config C4 in miniature: 6 views of one mesh, split 3 + 3 over two processes that share GPU 0 (gloo carries the
    300-KB-class vertex gradient): rank 0's face_idx / soft mask equal the single-process views 0..2 bit for bit, and the
    all-reduced vertex gradient equals the single-process gradient to 1e-5."""
    out = str(tmp_path / 'dibr.pt')
    mp.spawn(_dibr_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    import kaolin_amd as kal
    from kaolin_amd import distributed as D
    full = _dibr_views(kal, D)
    assert torch.equal(res['face_idx'], full['face_idx'][:3].cpu())
    assert torch.equal(res['soft'], full['soft'][:3].cpu())
    g_full, g = full['grad'].cpu().double(), res['grad'].double()
    assert float((g - g_full).abs().max()) <= 1e-5 * float(g_full.abs().max())


# ---- gradient accumulation over micro-batches: no_sync() (a second backward() must not reduce twice) ------
def _accum_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from kaolin_amd import distributed as D
    assert D.init_from_env('gloo')
    torch.manual_seed(2)
    verts = torch.rand(20, 3, dtype=torch.double, requires_grad=True)
    cams = torch.rand(8, 3, dtype=torch.double) + 2.0
    mine = D.shard_views(cams)                      # 4 views per rank, two micro-batches of 2
    reducer = D.SharedGradientReducer([verts])
    with reducer.no_sync():
        _view_loss(verts, mine[:2]).backward()
    assert reducer.posted == 0                      # accumulated locally, nothing on the wire
    _view_loss(verts, mine[2:]).backward()
    assert reducer.posted == 1                      # the accumulated sum, reduced once
    reducer.wait()
    reducer.remove()
    if rank == 0:
        torch.save({'verts': verts.grad}, out)
    D.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_micro_batches_reduce_once(tmp_path):
    out = str(tmp_path / 'accum.pt')
    mp.spawn(_accum_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    torch.manual_seed(2)
    verts = torch.rand(20, 3, dtype=torch.double, requires_grad=True)
    cams = torch.rand(8, 3, dtype=torch.double) + 2.0
    _view_loss(verts, cams).backward()
    assert torch.allclose(res['verts'], verts.grad, rtol=1e-12, atol=1e-14)


# ---- a world of ONE rank with the whole distributed control flow (KAMD_DIST_FORCE=1) ------------------------------------
_ONE_RANK_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ['KAMD_ROOT'])
import torch
import torch.distributed as dist
from kaolin_amd import distributed as D
assert D.init_from_env(), 'KAMD_DIST_FORCE=1 must initialise a process group for WORLD_SIZE=1'
assert D.is_distributed() and D.world_size() == 1 and D.rank() == 0
want = os.environ['KAMD_EXPECT_BACKEND']
assert dist.get_backend() == want, dist.get_backend()
dev = torch.device('cuda', torch.cuda.current_device()) if want == 'nccl' else torch.device('cpu')
torch.manual_seed(4)
verts = torch.rand(3000, 3, device=dev, requires_grad=True)
cams = torch.rand(6, 3, device=dev) + 2.0
reducer = D.SharedGradientReducer([verts])
steps = []
for step in range(3):                       # several steps: the Work objects of one step must not leak into the next
    verts.grad = None
    rel = verts.unsqueeze(0) - D.shard_views(cams).unsqueeze(1)
    ((rel[..., :2] / (1.0 + rel[..., 2:].abs())).pow(2).sum() + 0.1 * rel.norm(dim=-1).sum()).backward()
    assert reducer.posted == step + 1       # posted from autograd's hook, on the backward thread
    reducer.wait()                          # the current stream waits for the collective's stream
    steps.append(verts.grad.clone())
reducer.remove()
D.barrier()
if dev.type == 'cuda':
    torch.cuda.synchronize()
# all-reduce(SUM) over one rank is the identity: the gradient must equal the one computed without any collective
v2 = verts.detach().clone().requires_grad_()
rel = v2.unsqueeze(0) - cams.unsqueeze(1)
((rel[..., :2] / (1.0 + rel[..., 2:].abs())).pow(2).sum() + 0.1 * rel.norm(dim=-1).sum()).backward()
assert all(torch.equal(s, v2.grad) for s in steps)
t = torch.tensor([1.5], device=dev, dtype=torch.double)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
g = D.all_gather_batch(torch.ones(2, 1, device=dev))
assert float(t) == 1.5 and g.shape == (2, 1)
print(json.dumps({'ok': True, 'backend': dist.get_backend(), 'posted': reducer.posted}))
dist.destroy_process_group()
'''


def _run_one_rank(tmp_path, backend, extra_env=None):
    import subprocess
    script = tmp_path / 'one_rank_worker.py'
    script.write_text(_ONE_RANK_WORKER)
    env = dict(os.environ, KAMD_DIST_FORCE='1', KAMD_ROOT=ROOT, KAMD_EXPECT_BACKEND=backend, **(extra_env or {}))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert '"ok": true' in res.stdout
    return res.stdout


def test_forced_one_rank_world_gloo(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 1` with KAMD_DIST_FORCE=1: process group, hooked all-reduce,
    wait, barrier, all-gather all execute for a world of one rank (gloo on CPU here; RCCL below)."""
    _run_one_rank(tmp_path, 'gloo', {'KAMD_DIST_BACKEND': 'gloo', 'CUDA_VISIBLE_DEVICES': '', 'HIP_VISIBLE_DEVICES': ''})


@pytest.mark.gpu
def test_forced_one_rank_world_rccl(tmp_path):
    """The same over RCCL (backend "nccl") on the one GPU a box has: ncclAllReduce posted from autograd's
    post-accumulate hook with async_op=True, Work.wait() on the caller's stream, barrier, destroy -- the N > 1 control
    flow of bench.py and SharedGradientReducer on the backend the 8-GPU run uses."""
    out = _run_one_rank(tmp_path, 'nccl')
    assert '"backend": "nccl"' in out


@pytest.mark.gpu
def test_bench_under_torchrun_one_rank_rccl(tmp_path):
    """bench.py launched exactly as the driver launches it for N > 1 (torch.distributed.run, env rendezvous), one rank,
    KAMD_DIST_FORCE=1: init_process_group('nccl'), sharded views, the hooked vertex-gradient all-reduce inside the timed
    steps, the MAX-over-ranks reduction of the time, barrier and teardown all run on RCCL.  Small shapes (a smoke run)."""
    import json
    import subprocess
    env = dict(os.environ, KAMD_DIST_FORCE='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
           '--res', '256', '--sphere-frequency', '16', '--views-per-gpu', '4', '--chamfer-points', '20000', '--no-c5',
           '--no-cpu-baseline']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['distributed']['backend'] == 'nccl'
    assert line['distributed']['collectives_posted_per_step'] == 1


@pytest.mark.gpu
def test_bench_two_processes_sharing_gpu0(tmp_path):
    """bench.py's N > 1 path with TWO ranks (RCCL refuses two ranks on one device, so gloo carries the
    collectives between two processes that share GPU 0): launched as the driver launches it, `--gpus 2 --quick`.  The line
    must say n_gpus 2, 16 global views, one posted collective per step, and rank 0's all-reduced vertex gradient must equal
    the gradient of ONE process rendering the same 16 views (the MAX-over-ranks time reduction and the barriers run on the way)."""
    import json
    import subprocess
    common = ['--steps', '3', '--warmup', '1', '--res', '256', '--sphere-frequency', '16', '--quick']
    g2, g1 = str(tmp_path / 'grad2.pt'), str(tmp_path / 'grad1.pt')
    env = dict(os.environ, KAMD_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--views-per-gpu', '8',
           '--dump-vertex-grad', g2] + common
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=380)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['global_views'] == 16 and line['config']['views_per_gpu'] == 8
    assert line['distributed']['initialized'] and line['distributed']['backend'] == 'gloo'
    assert line['distributed']['collectives_posted_per_step'] == 1
    assert line['value'] > 0 and line['ms_per_step'] > 0 and line['scaling'] == 'weak'
    # value = the pixels of BOTH ranks over the slowest rank's time
    assert abs(line['value'] - 2 * 8 * 256 * 256 / (line['ms_per_step'] * 1e-3) / 1e6) <= 1e-3 * line['value']
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--views-per-gpu', '16', '--dump-vertex-grad', g1] + common,
                         env=dict(os.environ), capture_output=True, text=True, timeout=380)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    a, b = torch.load(g2).double(), torch.load(g1).double()
    assert a.shape == b.shape and float(b.abs().max()) > 0
    assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.gpu
def test_bench_gpus_flag_launches_its_own_ranks(tmp_path):
    """plain `python bench.py --gpus 2 --quick` (no launcher) must run TWO ranks by itself -- it re-executes
    under torch.distributed.run -- and print one line with n_gpus 2; a launcher-made job whose world disagrees with --gpus is an
    error, not a silent one-rank run."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['KAMD_DIST_BACKEND'] = 'gloo'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--quick', '--steps', '3', '--warmup', '1', '--res', '256',
           '--sphere-frequency', '16']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=380)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['config']['global_views'] == 16 and line['distributed']['initialized']
    assert line['distributed']['collectives_posted_per_step'] == 1 and line['distributed']['ranks_per_gpu'] == 2
    bad = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
                          '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--quick'],
                         env=env, capture_output=True, text=True, timeout=280)
    assert bad.returncode != 0 and '--gpus 2' in (bad.stdout + bad.stderr)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_eight_rank_dress_rehearsal_of_c4(tmp_path):
    """The 8-GPU run of config C4 as far as a one-GPU box can rehearse it: `python bench.py --gpus 8` launches eight ranks that
    share GPU 0 (gloo carries the collective), 8 views of the 50 000-face sphere at 1024^2 each = the 64 global views of
    BASELINE.json's configs[3]; the line must say so, post ONE collective per step, and rank 0's all-reduced vertex gradient must
    equal a single process rendering the same 64 views."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['KAMD_DIST_BACKEND'] = 'gloo'
    g8, g1 = str(tmp_path / 'grad8.pt'), str(tmp_path / 'grad1.pt')
    common = ['--quick', '--steps', '2', '--warmup', '1', '--no-scene-variants']
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--dump-vertex-grad', g8] + common, env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 8 and line['config']['global_views'] == 64 and line['config']['views_per_gpu'] == 8
    assert line['config']['height'] == 1024 and line['config']['faces'] == 50000
    assert line['distributed']['collectives_posted_per_step'] == 1 and line['scaling'] == 'weak'
    assert abs(line['value'] - 64 * 1024 * 1024 / (line['ms_per_step'] * 1e-3) / 1e6) <= 1e-3 * line['value']
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--views-per-gpu', '64', '--dump-vertex-grad', g1]
                         + common, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    a, b = torch.load(g8).double(), torch.load(g1).double()
    assert a.shape == b.shape and float(b.abs().max()) > 0
    assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
