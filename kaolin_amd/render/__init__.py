from . import camera  # noqa: F401
from . import mesh  # noqa: F401
