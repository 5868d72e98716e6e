from . import base, surrogate, functional, neuron, layer  # noqa: F401
