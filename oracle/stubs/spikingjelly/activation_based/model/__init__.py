from . import sew_resnet  # noqa: F401
