"""trianglemeshes_to_voxelgrids -- CPU restatement in numpy.  TEST INFRASTRUCTURE ONLY (see kaolin_oracle.c).

Restates the reference's pure-PyTorch algorithm:
  * kaolin/ops/conversions/trianglemesh.py:84-110  -- normalise (v - origin) / scale (defaults: per-mesh
    min and largest extent), per batch item subdivide then bin;
  * kaolin/ops/mesh/trianglemesh.py:410-458        -- repeat { keep triangles whose largest SQUARED edge
    length exceeds ((res-1)/res^2)^2 (compared in the tensor's dtype); midpoints v4=(v1+v3)/2,
    v5=(v1+v2)/2, v6=(v2+v3)/2; children (v1,v4,v5) (v2,v5,v6) (v4,v5,v6) (v3,v4,v6) } until none is kept;
    the result is the SET of original vertices and all generated midpoints (the reference de-duplicates
    with torch.unique each round, which does not change the set);
  * kaolin/ops/conversions/pointcloud.py:42-75     -- round(p * (res-1)) half-to-even, drop rows with a
    coordinate outside [0, res-1], mark the remaining voxels with 1 in a dense (res,res,res) grid.
Squared edge lengths are summed left to right (dx^2 + dy^2) + dz^2, which is what torch.sum gives for a
3-element reduction; all arithmetic is carried out in the input dtype.
"""
import numpy as np
import torch

_NP = {torch.float32: np.float32, torch.float64: np.float64, torch.float16: np.float16}


def _edge2(a, b):
    d = a - b
    sq = d * d
    return (sq[:, 0] + sq[:, 1]) + sq[:, 2]


def subdivide_points(vertices, faces, resolution):
    """(V,3) normalised vertices + (F,3) faces -> (P,3) array holding every original vertex and every
    midpoint the reference's subdivision generates (duplicates allowed: only the set matters)."""
    assert resolution > 1
    dt = vertices.dtype.type
    thr = dt(((resolution - 1) / (resolution ** 2)) ** 2)
    pts = [vertices]
    v1, v2, v3 = vertices[faces[:, 0]], vertices[faces[:, 1]], vertices[faces[:, 2]]
    two = dt(2)
    while v1.shape[0] > 0:
        longest = np.maximum(np.maximum(_edge2(v1, v2), _edge2(v2, v3)), _edge2(v3, v1))
        keep = longest > thr
        if not keep.any():
            break
        v1, v2, v3 = v1[keep], v2[keep], v3[keep]
        v4, v5, v6 = (v1 + v3) / two, (v1 + v2) / two, (v2 + v3) / two
        pts += [v4, v5, v6]
        v1, v2, v3 = (np.concatenate((v1, v2, v4, v3)), np.concatenate((v4, v5, v5, v4)),
                      np.concatenate((v5, v6, v6, v6)))
    return np.concatenate(pts)


def points_to_dense(points, resolution, dtype):
    idx = np.rint(points * points.dtype.type(resolution - 1)).astype(np.int64)
    ok = ((idx >= 0) & (idx <= resolution - 1)).all(axis=1)
    idx = idx[ok]
    grid = np.zeros((resolution,) * 3, dtype=dtype)
    grid[idx[:, 0], idx[:, 1], idx[:, 2]] = 1
    return grid


def trianglemeshes_to_voxelgrids(vertices, faces, resolution, origin=None, scale=None):
    """Dense result, same dtype as `vertices`; CPU torch tensors in and out."""
    if not isinstance(resolution, int):
        raise TypeError(f'Expected resolution to be int but got {type(resolution)}.')
    vertices = vertices.detach().cpu()
    faces_np = faces.detach().cpu().numpy()
    if origin is None:
        origin = torch.min(vertices, dim=1)[0]
    if scale is None:
        scale = torch.max(torch.max(vertices, dim=1)[0] - origin, dim=1)[0]
    normed = (vertices - origin.cpu().unsqueeze(1)) / scale.cpu().view(-1, 1, 1)   # same torch ops as the reference
    npdt = _NP[vertices.dtype]
    out = []
    for b in range(vertices.shape[0]):
        pts = subdivide_points(normed[b].numpy().astype(npdt, copy=False), faces_np, resolution)
        out.append(points_to_dense(pts, resolution, npdt))
    return torch.from_numpy(np.stack(out))
