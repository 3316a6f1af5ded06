"""Which part of the knot scene costs which kernel: the DIB-R step on sub-meshes (tube / + nested spheres / + bowl / all),
per-kernel averages from the library's own event instrumentation.  (GPU box; prints one line per sub-mesh.)"""
import math, sys, torch
sys.path.insert(0, '.')
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils import testing as T

dev = 'cuda'
H = W = 1024
V = 8
lib = _lib.load()
verts, faces = T.knot_mesh()
n_tube = 2 * 440 * 48
n_sph = 20 * 12 * 12 + 20 * 14 * 14
parts = {'tube': slice(0, n_tube), 'tube+spheres': slice(0, n_tube + n_sph),
         'tube+bowl': torch.cat([torch.arange(0, n_tube), torch.arange(n_tube + n_sph, faces.shape[0])]),
         'spheres+bowl': slice(n_tube, faces.shape[0]), 'all': slice(0, faces.shape[0]),
         'bowl': slice(n_tube + n_sph, faces.shape[0]),
         'all shuffled': torch.randperm(faces.shape[0], generator=torch.Generator().manual_seed(1)),
         'bowl x4 (subdivided)': None}
cams = T.fibonacci_cameras(V, 2.5).to(dev)
look_at = torch.zeros((V, 3), device=dev)
up = torch.tensor([[0., 1., 0.]], device=dev).repeat(V, 1)
rot, trans = kal.render.camera.generate_rotate_translate_matrices(cams, look_at, up)
proj = kal.render.camera.generate_perspective_projection(math.pi / 4).to(dev)
G1 = torch.rand((V, H, W, 3), device=dev)
G2 = torch.rand((V, H, W), device=dev)
for name, sel in parts.items():
    vsrc = verts
    if sel is None:      # every bowl face split into 4 (midpoints): half the size, 4x the count
        bf = faces[n_tube + n_sph:]
        a, b, c = verts[bf[:, 0]], verts[bf[:, 1]], verts[bf[:, 2]]
        ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
        tri = torch.stack([torch.stack(t, 1) for t in ((a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca))], 0).reshape(-1, 3, 3)
        vsrc = torch.cat([verts, tri.reshape(-1, 3)])
        extra = torch.arange(tri.shape[0] * 3).reshape(-1, 3) + verts.shape[0]
        f = torch.cat([faces[:n_tube + n_sph], extra]).to(dev).contiguous()
    else:
        f = faces[sel].to(dev).contiguous()
    v = vsrc.float().to(dev).requires_grad_()
    feats = torch.rand((V, f.shape[0], 3, 3), device=dev)

    def step():
        v.grad = None
        fv_cam, fv_img, normals = kal.render.mesh.prepare_vertices(v.unsqueeze(0).expand(V, -1, -1), f, proj, camera_rot=rot, camera_trans=trans)
        feat, soft, face_idx = kal.render.mesh.dibr_rasterization(H, W, fv_cam[..., 2], fv_img, feats, normals[..., 2])
        ((feat * G1).sum() + (soft * G2).sum()).backward()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    lib.kamd_profile_reset(); lib.kamd_profile_select(-1); lib.kamd_profile_enable(1)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    lib.kamd_profile_enable(0)
    prof = _lib.kernel_profile(reset=True)
    print('%-14s F %6d | ' % (name, f.shape[0]) + ' '.join('%s %.1f' % (k.replace('_kernel', ''), ms / n * 1e3) for k, (ms, n) in prof.items()), flush=True)
