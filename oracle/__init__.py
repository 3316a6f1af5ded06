"""Python access to the CPU oracle -- TEST INFRASTRUCTURE ONLY (see kaolin_oracle.c header).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  Nothing under ``kaolin_amd/`` imports it.

Functions take and return CPU ``torch`` tensors and mirror the signatures of the reference's
``kaolin._C`` operators.  ``build()`` compiles ``kaolin_oracle.c`` with gcc (serial + OpenMP).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def build(force=False):
    targets = ['libkaolin_oracle.so', 'libkaolin_oracle_omp.so']
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(('.c', '.inc'))]
    newest = max(os.path.getmtime(f) for f in srcs)
    stale = force or any(
        not os.path.exists(os.path.join(_HERE, t)) or
        os.path.getmtime(os.path.join(_HERE, t)) < newest for t in targets)
    if stale:
        subprocess.run(['make', '-C', _HERE, '-B'] if force else ['make', '-C', _HERE], check=True,
                       stdout=subprocess.DEVNULL)


def lib(omp=False):
    key = 'omp' if omp else 'serial'
    if key not in _libs:
        name = 'libkaolin_oracle_omp.so' if omp else 'libkaolin_oracle.so'
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        _libs[key] = ctypes.CDLL(path)
    return _libs[key]


def num_threads(omp=True):
    f = lib(omp).oracle_num_threads
    f.restype = ctypes.c_int
    return int(f())


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _cpu(t, dtype=None):
    t = t.detach().cpu().contiguous()
    if dtype is not None:
        t = t.to(dtype)
    return t


_SFX = {torch.float32: 'f32', torch.float64: 'f64', torch.float16: 'f16'}


# ---- sided distance -------------------------------------------------------------
def sided_distance_forward(p1, p2, omp=False):
    p1, p2 = _cpu(p1), _cpu(p2)
    B, N, M = p1.shape[0], p1.shape[1], p2.shape[1]
    dist = torch.zeros((B, N), dtype=p1.dtype)
    idx = torch.zeros((B, N), dtype=torch.long)
    f = getattr(lib(omp), f'oracle_sided_distance_forward_{_SFX[p1.dtype]}')
    f(ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(M), _p(p1), _p(p2), _p(dist), _p(idx))
    return dist, idx


def sided_distance_backward(grad, p1, p2, idx):
    grad, p1, p2, idx = _cpu(grad), _cpu(p1), _cpu(p2), _cpu(idx)
    B, N, M = p1.shape[0], p1.shape[1], p2.shape[1]
    g1, g2 = torch.zeros_like(p1), torch.zeros_like(p2)
    f = getattr(lib(False), f'oracle_sided_distance_backward_{_SFX[p1.dtype]}')
    f(ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(M), _p(grad), _p(p1), _p(p2), _p(idx), _p(g1), _p(g2))
    return g1, g2


# ---- DIB-R: the four kernels (reference `_C.render.mesh.*` signatures) ---------------
def _ci(v):
    return ctypes.c_int(int(v))


def _cf(v):
    return ctypes.c_float(float(v))


def packed_rasterize_forward(height, width, face_vertices_z, face_vertices_image, face_bboxes, face_features,
                             first_idx_face_per_mesh, multiplier, eps, omp=False):
    """rasterization.cpp:49-104: allocates sel_idx = -1, weights = 0, interp = 0, runs K1."""
    z, img, bbox, feat = _cpu(face_vertices_z), _cpu(face_vertices_image), _cpu(face_bboxes), _cpu(face_features)
    first = _cpu(first_idx_face_per_mesh, torch.long)
    B, D = first.shape[0] - 1, feat.shape[-1]
    interp = torch.zeros((B, height, width, D), dtype=z.dtype)
    sel = torch.full((B, height, width), -1, dtype=torch.long)
    wts = torch.zeros((B, height, width, 3), dtype=z.dtype)
    f = getattr(lib(omp), f'oracle_packed_rasterize_forward_{_SFX[z.dtype]}')
    f(_ci(B), _ci(height), _ci(width), _ci(D), _p(z), _p(img), _p(bbox), _p(feat), _p(first), _cf(multiplier), _cf(eps),
      _p(interp), _p(sel), _p(wts))
    return interp, sel, wts


def set_num_threads(n):
    """OpenMP build only: the number of threads its parallel loops use (bench.py: every hardware thread of the host)."""
    f = lib(True).oracle_set_num_threads
    f.argtypes = [ctypes.c_int]
    f.restype = None
    f(int(n))


def rasterize_backward(grad, face_idx, weights, face_vertices_image, face_features, eps, return_abs=False, omp=False):
    """rasterization.cpp:106-168 (interpolated_features is accepted by the reference but never read).  ``return_abs``: also
    the per-element sum of the terms' MAGNITUDES (float64) -- the scale any float32 accumulation's rounding error lives on."""
    grad, face_idx, weights = _cpu(grad), _cpu(face_idx, torch.long), _cpu(weights)
    img, feat = _cpu(face_vertices_image), _cpu(face_features)
    B, H, W, D = grad.shape
    F = img.shape[1]
    g_img, g_feat = torch.zeros_like(img), torch.zeros_like(feat)
    # (omp: pixels in parallel, atomic double adds in unspecified order -- the CPU baseline's timing; the parity tests use the serial build)
    f = getattr(lib(omp), f'oracle_rasterize_backward_{_SFX[grad.dtype]}')
    abs_img = torch.zeros(img.shape, dtype=torch.double) if return_abs else None
    f(_ci(B), _ci(H), _ci(W), _ci(F), _ci(D), _p(grad), _p(face_idx), _p(weights), _p(img), _p(feat), _cf(eps),
      _p(g_img), _p(g_feat), _p(abs_img) if return_abs else ctypes.c_void_p(0))
    return (g_img, g_feat, abs_img) if return_abs else (g_img, g_feat)


def dibr_soft_mask_forward(face_vertices_image, face_large_bboxes, selected_face_idx, sigmainv, knum, multiplier,
                           omp=False):
    """dibr_soft_mask.cpp:48-108; face_vertices_image is already multiplied by `multiplier`."""
    img, bbox, sel = _cpu(face_vertices_image), _cpu(face_large_bboxes), _cpu(selected_face_idx, torch.long)
    B, F = img.shape[0], img.shape[1]
    H, W = sel.shape[1], sel.shape[2]
    soft = torch.zeros((B, H, W), dtype=img.dtype)
    prob = torch.zeros((B, H, W, knum), dtype=img.dtype)
    idx = torch.full((B, H, W, knum), -1, dtype=torch.long)
    typ = torch.zeros((B, H, W, knum), dtype=torch.uint8)
    f = getattr(lib(omp), f'oracle_dibr_soft_mask_forward_{_SFX[img.dtype]}')
    f(_ci(B), _ci(H), _ci(W), _ci(F), _ci(knum), _p(img), _p(bbox), _p(sel), _cf(sigmainv), _cf(multiplier),
      _p(soft), _p(prob), _p(idx), _p(typ))
    return soft, prob, idx, typ


def dibr_soft_mask_backward(grad_soft_mask, soft_mask, selected_face_idx, close_face_prob, close_face_idx,
                            close_face_dist_type, face_vertices_image, sigmainv, multiplier, return_abs=False, omp=False):
    """dibr_soft_mask.cpp:110-183; face_vertices_image already scaled.  ``return_abs``: as in :func:`rasterize_backward`."""
    g, soft, sel = _cpu(grad_soft_mask), _cpu(soft_mask), _cpu(selected_face_idx, torch.long)
    prob, idx, typ = _cpu(close_face_prob), _cpu(close_face_idx, torch.long), _cpu(close_face_dist_type, torch.uint8)
    img = _cpu(face_vertices_image)
    B, F = img.shape[0], img.shape[1]
    H, W, K = sel.shape[1], sel.shape[2], prob.shape[-1]
    g_img = torch.zeros_like(img)
    f = getattr(lib(omp), f'oracle_dibr_soft_mask_backward_{_SFX[img.dtype]}')
    abs_img = torch.zeros(img.shape, dtype=torch.double) if return_abs else None
    f(_ci(B), _ci(H), _ci(W), _ci(F), _ci(K), _p(g), _p(soft), _p(sel), _p(prob), _p(idx), _p(typ), _p(img),
      _cf(sigmainv), _cf(multiplier), _p(g_img), _p(abs_img) if return_abs else ctypes.c_void_p(0))
    return (g_img, abs_img) if return_abs else g_img


# ---- DIB-R: the Python layers above the kernels (CPU torch, restating the reference's glue) ----
def rasterize(height, width, face_vertices_z, face_vertices_image, face_features, valid_faces=None,
              multiplier=1000, eps=1e-8, omp=False):
    """RasterizeCuda.forward (kaolin/render/mesh/rasterization.py:273-352): pack the valid faces
    (:292-317), scale by multiplier (:320), per-face bbox (:325-327), K1, map the packed index back to
    the mesh index and restore -1 (:340-346).  Returns (features, face_idx, weights)."""
    z, img, feat = _cpu(face_vertices_z), _cpu(face_vertices_image), _cpu(face_features)
    B, F = z.shape[0], z.shape[1]
    if valid_faces is None:
        valid = torch.ones((B, F), dtype=torch.bool)
    else:
        valid = _cpu(valid_faces).bool()
    bi, fi = torch.where(valid)
    first = torch.zeros(B + 1, dtype=torch.long)
    first[1:] = torch.cumsum(valid.reshape(B, -1).sum(dim=1), dim=0)
    pimg = img[bi, fi] * multiplier
    bbox = torch.cat((pimg.min(dim=1)[0], pimg.max(dim=1)[0]), dim=1)
    interp, sel, wts = packed_rasterize_forward(height, width, z[bi, fi], pimg, bbox, feat[bi, fi], first,
                                                multiplier, eps, omp=omp)
    if fi.numel() > 0:
        face_idx = fi[(sel + first[:-1].reshape(-1, 1, 1)).clamp(min=0).reshape(-1)].reshape(sel.shape).contiguous()
    else:
        face_idx = torch.full_like(sel, -1)
    face_idx[sel == -1] = -1
    return interp, face_idx, wts


def dibr_soft_mask(face_vertices_image, selected_face_idx, sigmainv=7000, boxlen=0.02, knum=30, multiplier=1000.,
                   omp=False):
    """DibrSoftMaskCuda.forward (kaolin/render/mesh/dibr.py:29-55). Returns (soft_mask, prob, idx, type,
    scaled_vertices)."""
    img = _cpu(face_vertices_image) * multiplier
    pmin, pmax = img.min(dim=-2)[0], img.max(dim=-2)[0]
    bbox = torch.cat([pmin - boxlen * multiplier, pmax + boxlen * multiplier], dim=-1)
    soft, prob, idx, typ = dibr_soft_mask_forward(img, bbox, selected_face_idx, sigmainv, knum, multiplier, omp=omp)
    return soft, prob, idx, typ, img


def dibr_rasterization(height, width, face_vertices_z, face_vertices_image, face_features, face_normals_z,
                       sigmainv=7000, boxlen=0.02, knum=30, multiplier=None, eps=None, omp=False):
    """dibr_rasterization (kaolin/render/mesh/dibr.py:119-209): rasterize with valid = normals_z >= 0,
    then the soft mask over ALL faces."""
    mult = 1000 if multiplier is None else multiplier
    e = 1e-8 if eps is None else eps
    feats, face_idx, wts = rasterize(height, width, face_vertices_z, face_vertices_image, face_features,
                                     _cpu(face_normals_z) >= 0., mult, e, omp=omp)
    soft, prob, idx, typ, simg = dibr_soft_mask(face_vertices_image, face_idx, sigmainv, boxlen, knum,
                                                1000. if multiplier is None else multiplier, omp=omp)
    return {'features': feats, 'face_idx': face_idx, 'weights': wts, 'soft_mask': soft, 'close_face_prob': prob,
            'close_face_idx': idx, 'close_face_dist_type': typ, 'scaled_vertices': simg}


# ---- point -> triangle-soup distance (reference `_C.metrics.unbatched_triangle_distance_*`) ----------
def triangle_distance_forward(points, face_vertices, omp=False, fused=True):
    """unbatched_triangle_distance.cpp:43-72 (outputs are caller-allocated there): returns (dist (N),
    face_idx (N) int64, dist_type (N) int32: 0 plane, 1-3 vertex, 4-6 edge).  ``fused=False``: the build without the
    contraction pin (every product and sum rounded on its own: tridist_oracle.inc header)."""
    pts, fv = _cpu(points), _cpu(face_vertices)
    N, F = pts.shape[0], fv.shape[0]
    dist = torch.zeros(N, dtype=pts.dtype)
    idx = torch.zeros(N, dtype=torch.long)
    typ = torch.zeros(N, dtype=torch.int32)
    f = getattr(lib(omp), f'oracle_triangle_distance_forward_{_SFX[pts.dtype]}' + ('' if fused else '_unfused'))
    f(_ci(N), _ci(F), _p(pts), _p(fv), _p(dist), _p(idx), _p(typ))
    return dist, idx, typ


def triangle_distance_backward(grad_dist, points, face_vertices, face_idx, dist_type, fused=True):
    """unbatched_triangle_distance.cpp:74-114 -> (grad_points (N,3), grad_face_vertices (F,3,3))."""
    g, pts, fv = _cpu(grad_dist), _cpu(points), _cpu(face_vertices)
    idx, typ = _cpu(face_idx, torch.long), _cpu(dist_type, torch.int32)
    N, F = pts.shape[0], fv.shape[0]
    gp, gf = torch.zeros_like(pts), torch.zeros_like(fv)
    f = getattr(lib(False), f'oracle_triangle_distance_backward_{_SFX[pts.dtype]}' + ('' if fused else '_unfused'))
    f(_ci(N), _ci(F), _p(g), _p(pts), _p(fv), _p(idx), _p(typ), _p(gp), _p(gf))
    return gp, gf


def point_to_mesh_distance(pointclouds, face_vertices, omp=False):
    """kaolin/metrics/trianglemesh.py:20-99: per-batch loop, stacked results."""
    out = [triangle_distance_forward(pointclouds[i], face_vertices[i], omp=omp) for i in range(pointclouds.shape[0])]
    return tuple(torch.stack([o[k] for o in out], dim=0) for k in range(3))


# ---- check_sign (ray-parity inside test; reference `_C.ops.mesh.unbatched_mesh_intersection_cuda`) ---------------
def mesh_intersection(points, verts_1, verts_2, verts_3, omp=False):
    """kaolin/csrc/ops/mesh/mesh_intersection.cpp: number of faces the +x ray of every point crosses -> (N) float."""
    pts, a, b, c = _cpu(points), _cpu(verts_1), _cpu(verts_2), _cpu(verts_3)
    N, F = pts.shape[0], a.shape[0]
    out = torch.zeros(N, dtype=pts.dtype)
    f = getattr(lib(omp), f'oracle_mesh_intersection_{_SFX[pts.dtype]}')
    f(_ci(N), _ci(F), _p(pts), _p(a), _p(b), _p(c), _p(out))
    return out


def check_sign(verts, faces, points, omp=False):
    """kaolin/ops/mesh/check_sign.py:45-155 (GPU branch): normalise by the largest extent, count, parity."""
    verts, faces, points = _cpu(verts), _cpu(faces, torch.long), _cpu(points)
    xlen = verts[..., 0].max(-1)[0] - verts[..., 0].min(-1)[0]
    ylen = verts[..., 1].max(-1)[0] - verts[..., 1].min(-1)[0]
    zlen = verts[..., 2].max(-1)[0] - verts[..., 2].min(-1)[0]
    maxlen = torch.max(torch.stack([xlen, ylen, zlen]), 0)[0]
    verts = verts / maxlen.view(-1, 1, 1)
    points = points / maxlen.view(-1, 1, 1)
    res = []
    for i in range(verts.shape[0]):
        v = verts[i]
        ints = mesh_intersection(points[i], v[faces[:, 0]], v[faces[:, 1]], v[faces[:, 2]], omp=omp)
        res.append(ints % 2 == 1.)
    return torch.stack(res)


# ---- deftet sparse render (multi-hit rasterization of free pixel coordinates) ------------------
def deftet_sparse_render_forward(face_vertices_z, face_vertices_image, face_bboxes, pixel_coords, pixel_depth_ranges,
                                 knum, eps, omp=False):
    """deftet.cpp:47-108: -> face_idx (B,P,knum) int64, pixel_depths, w0, w1 (first knum hits in mesh order, unsorted)."""
    z, img, bb = _cpu(face_vertices_z), _cpu(face_vertices_image), _cpu(face_bboxes)
    pix, rng = _cpu(pixel_coords), _cpu(pixel_depth_ranges)
    B, F = z.shape[:2]
    P = pix.shape[1]
    face_idx = torch.empty(B, P, knum, dtype=torch.long)
    depth, w0, w1 = (torch.empty(B, P, knum, dtype=z.dtype) for _ in range(3))
    f = getattr(lib(omp), f'oracle_deftet_forward_{_SFX[z.dtype]}')
    f(_ci(B), _ci(F), _ci(P), _ci(knum), _p(z), _p(img), _p(bb), _p(pix), _p(rng), _cf(eps), _p(face_idx), _p(depth),
      _p(w0), _p(w1))
    return face_idx, depth, w0, w1


def deftet_sparse_render_backward(grad_interpolated_features, face_idx, weights, face_vertices_image, face_features,
                                  eps):
    """deftet.cpp:110-161 -> grad_face_vertices_image, grad_face_features."""
    grad, face_idx, weights = _cpu(grad_interpolated_features), _cpu(face_idx, torch.long), _cpu(weights)
    img, feat = _cpu(face_vertices_image), _cpu(face_features)
    B, P, K, D = grad.shape
    F = img.shape[1]
    g_img, g_feat = torch.zeros_like(img), torch.zeros_like(feat)
    f = getattr(lib(False), f'oracle_deftet_backward_{_SFX[grad.dtype]}')
    f(_ci(B), _ci(F), _ci(P), _ci(K), _ci(D), _p(grad), _p(face_idx), _p(weights), _p(img), _p(feat), _cf(eps),
      _p(g_img), _p(g_feat))
    return g_img, g_feat


def deftet_sparse_render(pixel_coords, render_ranges, face_vertices_z, face_vertices_image, face_features, knum=300,
                         eps=1e-8, omp=False):
    """kaolin/render/mesh/deftet.py:269-315 (DeftetSparseRenderer.forward): boxes, the forward operator, sort by depth
    (descending; the reference's `torch.argsort` leaves the order of equal depths unspecified -- here, and in the
    product, equal depths keep mesh order), w2 = [hit] - (w0 + w1), corner features weighted and summed.
    -> dict(features (B,P,knum,D), face_idx, weights (B,P,knum,3))."""
    z, img, feat = _cpu(face_vertices_z), _cpu(face_vertices_image), _cpu(face_features)
    bb = torch.cat([img.min(dim=2)[0], img.max(dim=2)[0]], dim=2)
    face_idx, depth, w0, w1 = deftet_sparse_render_forward(z, img, bb, pixel_coords, render_ranges, knum, eps, omp=omp)
    order = torch.argsort(depth, descending=True, dim=-1, stable=True)
    face_idx = torch.gather(face_idx, -1, order).contiguous()
    w0, w1 = torch.gather(w0, -1, order), torch.gather(w1, -1, order)
    w2 = (face_idx != -1).to(z.dtype) - (w0 + w1)
    weights = torch.stack([w0, w1, w2], dim=-1).contiguous()
    B, F, _, D = feat.shape
    padded = torch.cat([torch.zeros(B, 1, 3, D, dtype=feat.dtype), feat], dim=1)
    sel = padded[torch.arange(B).view(B, 1, 1), face_idx + 1]           # (B,P,knum,3,D)
    features = (weights[..., 0, None] * sel[..., 0, :] + weights[..., 1, None] * sel[..., 1, :]) + \
        weights[..., 2, None] * sel[..., 2, :]
    return dict(features=features.contiguous(), face_idx=face_idx, weights=weights)


# ---- unbatched_mesh_to_spc (conservative voxelization into an SPC octree) ------------------------
def mesh_to_spc(face_vertices, level, omp=False, return_mortons=False):
    """kaolin/csrc/ops/conversions/mesh_to_spc/mesh_to_spc.cpp:27-42 -> (octree uint8, face_ids int64, barycoords (n,2) float)
    [+ the Morton codes of the occupied voxels]; the empty result is (0,), (0,), (0, 3) as in the reference (:347-351)."""
    fv = _cpu(face_vertices, torch.float32)
    nb = ctypes.c_int64(0)
    f = lib(omp).oracle_mesh_to_spc
    f.restype = ctypes.c_int64
    n = int(f(_ci(fv.shape[0]), _p(fv), _ci(level), ctypes.byref(nb)))
    if n == 0:
        out = (torch.empty(0, dtype=torch.uint8), torch.empty(0, dtype=torch.long), torch.zeros(0, 3))
        return out + (torch.empty(0, dtype=torch.long),) if return_mortons else out
    octree = torch.empty(nb.value, dtype=torch.uint8)
    face_ids, mortons = torch.empty(n, dtype=torch.long), torch.empty(n, dtype=torch.long)
    bary = torch.empty(n, 2, dtype=torch.float32)
    lib(omp).oracle_mesh_to_spc_fetch(_p(octree), _p(face_ids), _p(bary), _p(mortons))
    return (octree, face_ids, bary, mortons) if return_mortons else (octree, face_ids, bary)
