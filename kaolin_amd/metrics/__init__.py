from . import pointcloud  # noqa: F401
from . import trianglemesh  # noqa: F401
from . import render  # noqa: F401
