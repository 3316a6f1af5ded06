"""Generates the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container (where /root/reference is mounted):
    python tests/golden/make_golden.py [section ...]
The fixtures (.npz) are committed; this script is the record of how they were made.  Each section
imports the reference's own pure-PyTorch oracle by path and stores seeded inputs
plus the oracle's outputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload  # noqa: E402

SECTIONS = {}


def section(f):
    SECTIONS[f.__name__] = f
    return f


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name + '.npz', {k: v.shape for k, v in out.items()})


@section
def sided_distance():
    """reference oracle: kaolin/metrics/pointcloud.py:186-197 (_sided_distance, values only) +
    an fp64 brute-force argmin for the indices (unique minima on continuous random data)."""
    ref = _refload.load_reference()['pointcloud']
    cases = {}
    for tag, (B, N, M, scale, seed) in {'a': (3, 50, 50, 100., 0), 'b': (2, 257, 1031, 1., 1),
                                       'c': (1, 2000, 2000, 1., 0)}.items():
        torch.manual_seed(seed)
        if tag == 'c':   # BASELINE config C1: uniform [0,1)^3, seed 0, consecutive draws
            p1 = torch.rand(B, N, 3)
            p2 = torch.rand(B, M, 3)
        else:
            p1 = torch.randn(B, N, 3) * scale
            p2 = torch.randn(B, M, 3) * scale
        for dt, dn in ((torch.float32, 'f32'), (torch.float64, 'f64')):
            a, b = p1.to(dt), p2.to(dt)
            d = ref._sided_distance(a, b)
            dd = ((a.double()[:, :, None, :] - b.double()[:, None, :, :]) ** 2).sum(-1)
            idx = dd.argmin(-1)
            cases[f'{tag}_{dn}_dist'] = d
            cases[f'{tag}_{dn}_idx'] = idx
        cases[f'{tag}_p1'] = p1
        cases[f'{tag}_p2'] = p2
    save('sided_distance', **cases)




# --------------------------------------------------------------------------- DIB-R
def _model_obj_scene(dtype, batch_size=3, flip=False):
    """Inputs of the reference's DIB-R tests (tests/python/kaolin/render/mesh/test_rasterization.py:38-119,
    test_dibr.py:201-263): tests/samples/model.obj normalised to [0,1]^3, 3 cameras, fov pi/4, built with
    the REFERENCE's own camera functions."""
    import math
    sys.path.insert(0, os.path.join(HERE, os.pardir, os.pardir))
    from kaolin_amd.io import obj as myobj            # plain v / vt / f parser (the data is the reference's)
    from kaolin_amd.ops.mesh import index_vertices_by_faces
    cam = _refload.load_reference()['legacy_camera']
    mesh = myobj.import_mesh(os.path.join(_refload.REF, 'tests/samples/model.obj'))
    faces, face_uvs_idx = mesh.faces, mesh.face_uvs_idx
    if flip:
        faces, face_uvs_idx = torch.flip(faces, dims=(-1,)), torch.flip(face_uvs_idx, dims=(-1,))
    camera_pos = torch.tensor([[0.5, 0.5, 3.], [2., 2., -2.], [3., 0.5, 0.5]], dtype=dtype)[:batch_size]
    look_at = torch.full((batch_size, 3), 0.5, dtype=dtype)
    camera_up = torch.tensor([[0., 1., 0.]], dtype=dtype).repeat(batch_size, 1)
    proj = cam.generate_perspective_projection(fovyangle=math.pi / 4., dtype=dtype)
    v = mesh.vertices.to(dtype).unsqueeze(0)
    vmin, vmax = v.min(dim=1, keepdim=True)[0], v.max(dim=1, keepdim=True)[0]
    v = (v - vmin) / (vmax - vmin)
    rot, trans = cam.generate_rotate_translate_matrices(camera_pos, look_at, camera_up)
    v_cam = cam.rotate_translate_points(v, rot, trans)
    v_img = cam.perspective_camera(v_cam, proj)
    fz = index_vertices_by_faces(v_cam[:, :, -1:], faces).squeeze(-1)
    fimg = index_vertices_by_faces(v_img, faces)
    fuv = index_vertices_by_faces(mesh.uvs.unsqueeze(0).to(dtype), face_uvs_idx).repeat(batch_size, 1, 1, 1)
    zmin = fz.reshape(batch_size, -1).min(dim=1, keepdim=True)[0]
    zmax = fz.reshape(batch_size, -1).max(dim=1, keepdim=True)[0]
    valid = torch.all(fz < ((zmin + zmax) / 2.).unsqueeze(-1), dim=-1)
    rr = torch.stack([v_cam[:, :, -1].min(dim=1)[0] - 1e-2, v_cam[:, :, -1].max(dim=1)[0] + 1e-2], dim=-1)
    return fz, fimg, fuv, valid, rr


@section
def rasterize():
    """reference oracle: _naive_deftet_sparse_render(knum=1) (kaolin/render/mesh/deftet.py:101-267) on the
    fixtures of test_rasterization.py:137-158 (32x32, model.obj, 3 cameras, flip, valid_faces)."""
    deftet = _refload.load_reference()['deftet']
    H = W = 32
    out = {}
    for dn, dtype in (('f32', torch.float), ('f64', torch.double)):
        for flip in (False, True):
            fz, fimg, fuv, valid, rr = _model_obj_scene(dtype, 3, flip)
            x = (2 * torch.arange(W, dtype=dtype) + 1 - W) / W
            y = (H - 2 * torch.arange(H, dtype=dtype) - 1.) / H
            pix = torch.stack([x.reshape(1, 1, -1).repeat(3, H, 1), y.reshape(1, -1, 1).repeat(3, 1, W)],
                              dim=-1).reshape(3, -1, 2)
            ranges = rr.unsqueeze(1).repeat(1, H * W, 1)
            tag = f'{dn}_flip{int(flip)}'
            out[tag + '_z'], out[tag + '_img'], out[tag + '_uv'], out[tag + '_valid'] = fz, fimg, fuv, valid
            for wv in (False, True):
                kw = {'valid_faces': valid} if wv else {}
                feats, idx = deftet._naive_deftet_sparse_render(pix, ranges, fz, fimg, fuv, 1, **kw)
                out[f'{tag}_valid{int(wv)}_face_idx'] = idx.reshape(3, H, W).to(torch.int32)
                out[f'{tag}_valid{int(wv)}_feat'] = feats.reshape(3, H, W, 2)
    save('rasterize', **out)


@section
def rasterize_backward():
    """Autograd through the reference's torch oracle (_naive_deftet_sparse_render, knum=1) with a seeded
    grad_out, as test_rasterization.py:190-233 does; float64 only (the reference compares at rtol 1e-3 /
    atol 1e-2 (vertices), 1e-3 (features))."""
    deftet = _refload.load_reference()['deftet']
    H = W = 32
    dtype = torch.double
    out = {}
    for flip in (False, True):
        fz, fimg, fuv, valid, rr = _model_obj_scene(dtype, 3, flip)
        x = (2 * torch.arange(W, dtype=dtype) + 1 - W) / W
        y = (H - 2 * torch.arange(H, dtype=dtype) - 1.) / H
        pix = torch.stack([x.reshape(1, 1, -1).repeat(3, H, 1), y.reshape(1, -1, 1).repeat(3, 1, W)],
                          dim=-1).reshape(3, -1, 2)
        ranges = rr.unsqueeze(1).repeat(1, H * W, 1)
        a = fimg.clone().requires_grad_()
        u = fuv.clone().requires_grad_()
        feats, idx = deftet._naive_deftet_sparse_render(pix, ranges, fz, a, u, 1)
        torch.manual_seed(7)
        grad_out = torch.rand(3, H, W, 2, dtype=dtype)
        feats.reshape(3, H, W, 2).backward(grad_out)
        tag = f'flip{int(flip)}'
        out[tag + '_grad_out'], out[tag + '_g_img'], out[tag + '_g_uv'] = grad_out, a.grad, u.grad
    save('rasterize_backward', **out)


@section
def triangle_distance():
    """reference oracle: _unbatched_naive_point_to_mesh_distance (kaolin/metrics/trianglemesh.py:151-276) on
    seeded random points/faces (shape of test_trianglemesh.py:81-122: 1025 x 1025, crossing the 1024/512 face
    tile) and on a small sphere mesh; gradients by autograd through the oracle with a seeded grad_out."""
    tm = _refload.load_reference()['trianglemesh']
    sys.path.insert(0, os.path.join(HERE, os.pardir, os.pardir))
    from kaolin_amd.utils.testing import geodesic_sphere
    out = {}
    for tag in ('rand', 'sphere'):
        for dn, dtype in (('f32', torch.float), ('f64', torch.double)):
            torch.manual_seed(0)
            if tag == 'rand':
                pts = torch.randn(1025, 3, dtype=dtype)
                fv = torch.randn(1025, 3, 3, dtype=dtype)
            else:
                v, f = geodesic_sphere(4)
                fv = v.to(dtype)[f]
                pts = torch.rand(700, 3, dtype=dtype) * 1.4 - 0.7
            a, b = pts.clone().requires_grad_(), fv.clone().requires_grad_()
            dist, idx, typ = tm._unbatched_naive_point_to_mesh_distance(a, b)
            g = torch.rand(dist.shape, dtype=dtype)
            dist.backward(g)
            k = f'{tag}_{dn}'
            out[k + '_points'], out[k + '_faces'], out[k + '_grad_out'] = pts, fv, g
            out[k + '_dist'], out[k + '_idx'], out[k + '_type'] = dist, idx, typ.to(torch.int32)
            out[k + '_g_points'], out[k + '_g_faces'] = a.grad, b.grad
    save('triangle_distance', **out)


@section
def voxelgrid():
    """reference: trianglemeshes_to_voxelgrids (kaolin/ops/conversions/trianglemesh.py:29-110, pure torch, run
    here on CPU) on the inputs of its own tests (tests/python/kaolin/ops/conversions/test_trianglemesh.py:45-242)
    and on seeded random / sphere meshes; dense outputs stored bit-packed."""
    conv = _refload.load_reference()['conv_trianglemesh'].trianglemeshes_to_voxelgrids
    sys.path.insert(0, os.path.join(HERE, os.pardir, os.pardir))
    from kaolin_amd.utils.testing import geodesic_sphere
    out = {}
    cases = []
    tri = torch.tensor([[[0, 0, 0], [1, 0, 0], [0, 0, 1]], [[0, 0, 0], [0, 1, 0], [1, 0, 1]]], dtype=torch.float)
    f1 = torch.tensor([[0, 1, 2]])
    cases.append(('batched', tri, f1, 3, torch.zeros(2, 3), torch.ones(2)))
    cases.append(('origins', tri[:1], f1, 3, torch.tensor([[0., 0.5, 0.]]), torch.ones(1)))      # shape of :93-123
    cases.append(('scale', tri[:1], f1, 3, torch.zeros(1, 3), torch.ones(1) * 2))                 # shape of :125-154
    cases.append(('res4', tri[:1], f1, 4, None, None))
    rect_v = torch.tensor([[[0, 0, 0], [8, 0, 0], [0, 8, 0], [8, 8, 0], [0, 0, 12], [8, 0, 12], [0, 8, 12], [8, 8, 12]]],
                          dtype=torch.float)
    rect_f = torch.tensor([[0, 3, 1], [0, 2, 3], [0, 1, 5], [0, 5, 4], [6, 7, 3], [6, 3, 2], [1, 3, 7], [1, 7, 5],
                           [4, 5, 7], [4, 7, 6], [4, 6, 2], [4, 2, 0]])
    cases.append(('rect', rect_v, rect_f, 16, None, None))
    torch.manual_seed(0)
    cases.append(('rand', torch.rand(2, 40, 3), torch.randint(0, 40, (60, 3)), 32, None, None))
    cases.append(('rand_out', torch.rand(1, 30, 3) * 2 - 0.5, torch.randint(0, 30, (40, 3)), 24, torch.zeros(1, 3), torch.ones(1)))
    v, f = geodesic_sphere(4)
    cases.append(('sphere64', v.float()[None], f, 64, None, None))
    cases.append(('sphere40_f64', v[None], f, 40, None, None))
    for name, vv, ff, res, org, sc in cases:
        dense = conv(vv, ff, res, org, sc, False)
        out[name + '_vertices'], out[name + '_faces'], out[name + '_res'] = vv, ff, res
        if org is not None:
            out[name + '_origin'], out[name + '_scale'] = org, sc
        out[name + '_packed'] = np.packbits(dense.numpy().astype(np.uint8).reshape(-1))
        out[name + '_count'] = int(dense.sum())
    save('voxelgrid', **out)


def _dibr_gt(sub, stem, H, W, sigmainv, boxlen):
    return torch.load(os.path.join(_refload.REF, 'tests/samples/dibr', sub, f'{stem}_{H}_{W}_{int(sigmainv)}_{boxlen}.pt'),
                      map_location='cpu')


@section
def dibr_soft_mask():
    """Golden tensors produced by the reference's CUDA kernels (Kaolin v0.10.0), shipped in
    tests/samples/dibr/{simple,sphere}/*.pt and used by test_dibr.py:41-394; repacked losslessly
    (float64 kept, idx -> int16 0-based, type -> uint8) together with the sphere inputs built by
    the reference's camera code."""
    H, W = 35, 31
    out = {}
    for sub, boxlens in (('simple', (0.02, 0.2)), ('sphere', (0.02, 0.01))):
        for sigmainv in (7000, 70):
            for boxlen in boxlens:
                tag = f'{sub}_{sigmainv}_{boxlen}'
                out[tag + '_soft_mask'] = _dibr_gt(sub, 'soft_mask', H, W, sigmainv, boxlen)
                out[tag + '_idx'] = (_dibr_gt(sub, 'close_face_idx', H, W, sigmainv, boxlen).long() - 1).to(torch.int16)
                out[tag + '_prob'] = _dibr_gt(sub, 'close_face_dist', H, W, sigmainv, boxlen)
                out[tag + '_type'] = _dibr_gt(sub, 'close_face_dist_type', H, W, sigmainv, boxlen).to(torch.uint8)
                out[tag + '_grad'] = _dibr_gt(sub, 'grad_face_vertices_image', H, W, sigmainv, boxlen)
    for dn, dtype in (('f32', torch.float), ('f64', torch.double)):
        for flip in (False, True):
            fz, fimg, _, _, _ = _model_obj_scene(dtype, 3, flip)
            out[f'sphere_in_{dn}_flip{int(flip)}_z'] = fz
            out[f'sphere_in_{dn}_flip{int(flip)}_img'] = fimg
    out['simple_new_face_idx'] = torch.load(os.path.join(_refload.REF, 'tests/samples/dibr/simple/new_face_idx_35_31.pt'),
                                            map_location='cpu').to(torch.int16)
    save('dibr_soft_mask', **out)


@section
def deftet():
    """reference oracle: _naive_deftet_sparse_render (kaolin/render/mesh/deftet.py:101-267) on the fixtures of
    tests/python/kaolin/render/mesh/test_deftet.py:332-441 (model.obj, 3 cameras, seeded random pixel coordinates in
    [-1,1]^2, render range [zmin | (zmin+zmax)/2, 0], knum 20): face_idx + features for float / double, and for double the
    autograd gradients of a seeded grad_out (test_backward, :443-489).  hits per pixel stay below knum, so "first knum
    in mesh order, then sorted" (the CUDA operator) and "first knum by depth" (this oracle) coincide."""
    dt = _refload.load_reference()['deftet']
    P, knum = 257, 20
    out = {}
    for dn, dtype in (('f32', torch.float), ('f64', torch.double)):
        fz, fimg, fuv, valid, rr = _model_obj_scene(dtype, 3, False)
        torch.manual_seed(11)
        pix = torch.rand(3, P, 2, dtype=dtype) * 2. - 1.
        zmin, zmax = fz.reshape(3, -1).min(dim=1)[0], fz.reshape(3, -1).max(dim=1)[0]
        out[f'{dn}_z'], out[f'{dn}_img'], out[f'{dn}_uv'], out[f'{dn}_pix'] = fz, fimg, fuv, pix
        for centre in (0, 1):
            lo = (zmin + zmax) / 2. if centre else zmin
            ranges = torch.nn.functional.pad(lo.unsqueeze(-1), (0, 1), value=0.).unsqueeze(1).repeat(1, P, 1)
            tag = f'{dn}_centre{centre}'
            out[tag + '_ranges'] = ranges
            a, u = fimg.clone().requires_grad_(), fuv.clone().requires_grad_()
            feats, idx = dt._naive_deftet_sparse_render(pix, ranges, fz, a, u, knum)
            assert int((idx != -1).sum(-1).max()) < knum
            out[tag + '_face_idx'], out[tag + '_feat'] = idx.to(torch.int32), feats.detach()
            if dtype == torch.double:
                torch.manual_seed(13 + centre)
                grad_out = torch.rand_like(feats)
                feats.backward(grad_out)
                out[tag + '_grad_out'], out[tag + '_g_img'], out[tag + '_g_uv'] = grad_out, a.grad, u.grad
    save('deftet', **out)


if __name__ == '__main__':
    todo = sys.argv[1:] or list(SECTIONS)
    for s in todo:
        SECTIONS[s]()
