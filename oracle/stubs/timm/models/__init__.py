from . import layers  # noqa: F401
