from . import mesh  # noqa: F401
from . import conversions  # noqa: F401
