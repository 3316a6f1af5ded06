"""Minimal Wavefront OBJ reader: triangles with `v`, `vt`, `f a/b/c` records -- what the hot-path
fixtures need (the reference's full importer, kaolin/io/obj.py, is out of scope: SURVEY.md section 2)."""
from collections import namedtuple

import torch

ObjMesh = namedtuple('ObjMesh', ['vertices', 'faces', 'uvs', 'face_uvs_idx'])


def import_mesh(path):
    verts, uvs, faces, face_uvs = [], [], [], []
    with open(path) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == 'v':
                verts.append([float(x) for x in tok[1:4]])
            elif tok[0] == 'vt':
                uvs.append([float(x) for x in tok[1:3]])
            elif tok[0] == 'f':
                refs = [t.split('/') for t in tok[1:]]
                if len(refs) != 3:
                    raise ValueError('only triangle meshes are supported')
                faces.append([int(r[0]) - 1 for r in refs])
                if len(refs[0]) > 1 and refs[0][1] != '':
                    face_uvs.append([int(r[1]) - 1 for r in refs])
    return ObjMesh(torch.tensor(verts, dtype=torch.float), torch.tensor(faces, dtype=torch.long),
                   torch.tensor(uvs, dtype=torch.float) if uvs else None,
                   torch.tensor(face_uvs, dtype=torch.long) if face_uvs else None)
