"""Development probe: the C4 step as bench.py runs it (prepare_vertices -> dibr_rasterization -> fused weighted_sum loss -> backward,
static face features) with the library's per-kernel events: kernel us per step.  usage: python tools/round6/step_kernels.py [N] [scene]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils import testing as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
scene = sys.argv[2] if len(sys.argv) > 2 else 'sphere'
V, H, W = 8, 1024, 1024
dev = 'cuda'
v, f = T.scene_mesh(scene, 50) if hasattr(T, 'scene_mesh') else T.geodesic_sphere(50)
verts = v.float().to(dev).requires_grad_()
faces = f.to(dev)
cams = T.fibonacci_cameras(V, 2.5).to(dev)   # (bench.py build_scene at N = 1)
rot, trans = kal.render.camera.generate_rotate_translate_matrices(cams, torch.zeros_like(cams), torch.tensor([[0., 1., 0.]], device=dev).expand(V, -1))
proj = kal.render.camera.generate_perspective_projection(math.pi / 4).to(dev)
g = torch.Generator().manual_seed(0)
feats3 = torch.cat([torch.rand((1, faces.shape[0], 3, 2), generator=g), torch.ones((1, faces.shape[0], 3, 1))], -1).to(dev).expand(V, -1, -1, -1).contiguous()
G1, G2 = torch.rand((V, H, W, 3), generator=g).to(dev), torch.rand((V, H, W), generator=g).to(dev)
lib = _lib.load()


def step():
    verts.grad = None
    fv_cam, fv_img, normals = kal.render.mesh.prepare_vertices(verts.unsqueeze(0).expand(V, -1, -1), faces, proj, camera_rot=rot, camera_trans=trans)
    feat, soft, face_idx = kal.render.mesh.dibr_rasterization(H, W, fv_cam[..., 2], fv_img, feats3, normals[..., 2])
    kal.metrics.render.weighted_sum(feat, G1, soft, G2).backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(n):
    step()
torch.cuda.synchronize()
plain = (time.time() - t0) / n
lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
for _ in range(n):
    step()
torch.cuda.synchronize()
lib.kamd_profile_enable(0)
print(scene, 'step ms', round(plain * 1e3, 4), {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()})
