"""Timing-only probe of the binning kernel (experiment builds that return after it): sub-meshes of the knot scene."""
import math, sys, torch
sys.path.insert(0, '.')
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils import testing as T
dev = 'cuda'; H = W = 1024; V = 8
lib = _lib.load()
verts, faces = T.knot_mesh()
n_tube = 2 * 440 * 48; n_sph = 20 * 12 * 12 + 20 * 14 * 14
bf = faces[n_tube + n_sph:]
def subdivide(verts, bf):
    a, b, c = verts[bf[:, 0]], verts[bf[:, 1]], verts[bf[:, 2]]
    ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
    tri = torch.stack([torch.stack(t, 1) for t in ((a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca))], 0).reshape(-1, 3, 3)
    return tri.reshape(-1, 3), torch.arange(tri.shape[0] * 3).reshape(-1, 3)
v1, f1 = subdivide(verts, bf)
v2, f2 = subdivide(v1, f1)
cases = {'tube': (verts, faces[:n_tube]), 'bowl': (verts, bf), 'bowl/4': (v1, f1), 'bowl/16': (v2, f2), 'all': (verts, faces)}
cams = T.fibonacci_cameras(V, 2.5).to(dev)
rot, trans = kal.render.camera.generate_rotate_translate_matrices(cams, torch.zeros((V, 3), device=dev), torch.tensor([[0., 1., 0.]], device=dev).repeat(V, 1))
proj = kal.render.camera.generate_perspective_projection(math.pi / 4).to(dev)
out = []
for name, (vv, ff) in cases.items():
    f = ff.to(dev).contiguous(); v = vv.float().to(dev)
    feats = torch.rand((V, f.shape[0], 3, 3), device=dev)
    with torch.no_grad():
        fv_cam, fv_img, normals = kal.render.mesh.prepare_vertices(v.unsqueeze(0).expand(V, -1, -1), f, proj, camera_rot=rot, camera_trans=trans)
        for _ in range(3):
            kal.render.mesh.dibr_rasterization(H, W, fv_cam[..., 2], fv_img, feats, normals[..., 2])
        torch.cuda.synchronize()
        lib.kamd_profile_reset(); lib.kamd_profile_select(-1); lib.kamd_profile_enable(1)
        for _ in range(10):
            kal.render.mesh.dibr_rasterization(H, W, fv_cam[..., 2], fv_img, feats, normals[..., 2])
        torch.cuda.synchronize()
        lib.kamd_profile_enable(0)
    prof = _lib.kernel_profile(reset=True)
    ms, n = prof['bin_faces_kernel']
    out.append('%s(F=%d) %.1f' % (name, f.shape[0], ms / n * 1e3))
print(' | '.join(out), flush=True)
