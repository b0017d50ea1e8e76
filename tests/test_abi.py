"""The C-ABI shared library loads and exports every symbol include/porefv.h declares; the
product refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

import porepy_amd as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "porefv.h")).read()
    names = set(re.findall(r"\b(pfv_[a-z_]+)\s*\(", text))
    return sorted(names - {"pfv_status"})  # "(pfv_status)" casts in comments are not symbols


def test_header_and_binding_agree():
    syms = declared_symbols()
    assert len(syms) >= 15
    assert sorted(pa._lib.EXPORTS) == syms


@pytest.mark.skipif(not os.path.exists(pa._lib.DEFAULT_LIBRARY), reason="product library not built yet")
def test_product_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(pa._lib.DEFAULT_LIBRARY)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    lib.pfv_is_device_build.restype = ctypes.c_int
    assert lib.pfv_is_device_build() == 1


@pytest.mark.skipif(not os.path.exists(pa._lib.DEFAULT_LIBRARY), reason="product library not built yet")
def test_product_fails_loudly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pa.PorefvError):
        pa.Context(0)
    g = pa.CartGrid([2, 2], [1, 1])
    g.compute_geometry()
    data = pa.initialize_data({}, "flow", {"second_order_tensor": pa.SecondOrderTensor([1.0] * 4),
                                           "bc": pa.BoundaryCondition(g)})
    with pytest.raises(pa.PorefvError):
        pa.Mpfa("flow").discretize(g, data)
