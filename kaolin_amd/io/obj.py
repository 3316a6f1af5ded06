"""Minimal Wavefront OBJ reader: triangles with ``v`` / ``vt`` / ``f a/b/c`` records and, on request, the materials
of the referenced ``.mtl`` file (colours and image maps) -- what the hot path's fixtures and the reference's rasterizer /
DIB-R tests need.  The reference's full importer (kaolin/io/obj.py: polygons, normals, error handlers, material
assignments, PBR conversion) is out of scope."""
import os
from collections import namedtuple

import torch

ObjMesh = namedtuple('ObjMesh', ['vertices', 'faces', 'uvs', 'face_uvs_idx', 'materials'])


def _load_image(path):
    """(H, W, C) uint8 tensor of an image file (first three channels)."""
    import numpy as np
    from PIL import Image
    img = np.array(Image.open(path))
    if img.ndim == 2:
        img = img[:, :, None]
    return torch.from_numpy(img[:, :, :3].copy())


def load_mtl(path):
    """``{material name: {'Ka' / 'Kd' / 'Ks': (3,) tensors, 'map_Ka' / 'map_Kd' / 'map_Ks': (H, W, 3) uint8 tensors}}``
    in file order (the subset of kaolin/io/obj.py:326-400 the tests read)."""
    mats, cur = {}, None
    folder = os.path.dirname(path)
    with open(path) as fh:
        for line in fh:
            tok = line.split()
            if not tok or tok[0].startswith('#'):
                continue
            if tok[0] == 'newmtl':
                cur = mats.setdefault(tok[1], {'material_name': tok[1]})
            elif cur is not None and tok[0] in ('Ka', 'Kd', 'Ks'):
                cur[tok[0]] = torch.tensor([float(x) for x in tok[1:4]])
            elif cur is not None and tok[0] in ('map_Ka', 'map_Kd', 'map_Ks'):
                cur[tok[0]] = _load_image(os.path.join(folder, tok[-1]))
    return mats


def import_mesh(path, with_materials=False):
    verts, uvs, faces, face_uvs, mtl_files = [], [], [], [], []
    with open(path) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == 'v':
                verts.append([float(x) for x in tok[1:4]])
            elif tok[0] == 'vt':
                uvs.append([float(x) for x in tok[1:3]])
            elif tok[0] == 'mtllib':
                mtl_files.append(tok[1])
            elif tok[0] == 'f':
                refs = [t.split('/') for t in tok[1:]]
                if len(refs) != 3:
                    raise ValueError('only triangle meshes are supported')
                faces.append([int(r[0]) - 1 for r in refs])
                if len(refs[0]) > 1 and refs[0][1] != '':
                    face_uvs.append([int(r[1]) - 1 for r in refs])
    materials = None
    if with_materials:
        materials = []
        for name in mtl_files:
            materials.extend(load_mtl(os.path.join(os.path.dirname(path), name)).values())
    return ObjMesh(torch.tensor(verts, dtype=torch.float), torch.tensor(faces, dtype=torch.long),
                   torch.tensor(uvs, dtype=torch.float) if uvs else None,
                   torch.tensor(face_uvs, dtype=torch.long) if face_uvs else None, materials)
