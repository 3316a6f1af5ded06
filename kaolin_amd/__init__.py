"""kaolin_amd -- MI355X-native (gfx950) implementation of Kaolin's DIB-R
differentiable-rasterization and 3D-metrics hot path behind Kaolin's own Python API.

    kaolin_amd.render.mesh      rasterize, dibr_soft_mask, dibr_rasterization (+ helpers)
    kaolin_amd.metrics          pointcloud.{sided_distance, chamfer_distance, f_score},
                                trianglemesh.point_to_mesh_distance, render.mask_iou
    kaolin_amd.ops.conversions  trianglemeshes_to_voxelgrids
    kaolin_amd._C               the 8 operator bindings of ``kaolin._C`` on this path
    kaolin_amd.distributed      batch/view sharding over RCCL (new; the reference has none)

``kaolin_amd.install_as_kaolin()`` registers the package under the name ``kaolin`` so that
existing notebooks/tests written against the reference import it unchanged.
"""
import sys

from . import _C  # noqa: F401
from . import io, metrics, ops, render, utils  # noqa: F401
from . import distributed  # noqa: F401

__version__ = '0.1.0'


def install_as_kaolin():
    """Alias this package (and the submodules on the hot path) as ``kaolin`` in sys.modules."""
    prefix = __name__ + '.'
    for name, mod in list(sys.modules.items()):
        if name == __name__:
            sys.modules['kaolin'] = mod
        elif name.startswith(prefix):
            sys.modules['kaolin.' + name[len(prefix):]] = mod
    return sys.modules['kaolin']
