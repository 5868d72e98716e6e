"""CPU-side checks of the C ABI: the library loads and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "sdformerflow_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|int64_t) (sdf_\w+)\(", src, flags=re.M)))


def test_header_declares_the_expected_entry_points():
    from sdformerflow_amd import hip
    assert set(declared_symbols()) == set(hip.EXPORTS)


def test_library_exports_every_declared_symbol():
    from sdformerflow_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    lib.sdf_version.restype = ctypes.c_int
    assert lib.sdf_version() == 100            # host-only call, no GPU needed


def test_product_refuses_cpu_tensors():
    import torch
    from sdformerflow_amd import hip
    with pytest.raises(hip.SdfError):
        hip.lif_fwd(torch.zeros(2, 8))
