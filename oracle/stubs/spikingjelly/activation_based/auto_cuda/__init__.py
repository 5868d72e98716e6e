from . import cfunction  # noqa: F401
