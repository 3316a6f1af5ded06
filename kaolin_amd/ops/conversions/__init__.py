from .trianglemesh import trianglemeshes_to_voxelgrids  # noqa: F401
from . import trianglemesh  # noqa: F401
