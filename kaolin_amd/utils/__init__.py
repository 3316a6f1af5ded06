from . import testing  # noqa: F401
