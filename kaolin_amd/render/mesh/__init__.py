from .rasterization import rasterize  # noqa: F401
from .dibr import dibr_soft_mask, dibr_rasterization  # noqa: F401
from .utils import prepare_vertices, texture_mapping  # noqa: F401
from .deftet import deftet_sparse_render  # noqa: F401
from . import rasterization, dibr, utils, deftet  # noqa: F401
