from .trianglemesh import trianglemeshes_to_voxelgrids, unbatched_mesh_to_spc  # noqa: F401
from . import trianglemesh  # noqa: F401
