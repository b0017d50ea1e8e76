"""GPU suite (-m gpu): the gfx950 HIP library driven through the C ABI, checked against the
oracle, the committed golden fixtures of the reference, and size-independent properties at
benchmark scale.  Nothing here reads /root/reference."""
import numpy as np
import pytest

import porepy_amd as pa
from tests import _parity as P
from tests._golden import case_names

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return pa._lib.product_library()  # raises if the HIP build is missing


def _geo(g):
    g.compute_geometry()
    return g


def test_device_build_loaded(lib):
    assert lib.pfv_is_device_build() == 1


@pytest.mark.parametrize("name", case_names())
def test_golden_case(lib, name):
    P.check_golden_case(lib, name)


def test_generic_pattern_bit_exact(lib):
    P.check_generic_pattern_bit_exact(lib)


@pytest.mark.parametrize("key", ["cart_homogeneous", "cart_heterogeneous",
                                 "simplex_homogeneous", "simplex_heterogeneous"])
def test_reference_known_answers(lib, key):
    P.check_known_answer(lib, key)


@pytest.mark.parametrize("make", [
    lambda: pa.CartGrid([30, 20], [1, 1]),
    lambda: pa.CartGrid([8, 7, 6], [1, 1, 1]),
    lambda: pa.perturb_interior_nodes(_geo(pa.StructuredTriangleGrid([15, 14], [1, 1])), 0.02),
    lambda: pa.perturb_interior_nodes(_geo(pa.StructuredTetrahedralGrid([8, 8, 8], [1, 1, 1])), 0.03),
])
def test_operator_api_vs_oracle(lib, make):
    g = make()
    g.compute_geometry()
    P.operator_roundtrip(lib, g)


def test_heterogeneous_1e6(lib):
    g = _geo(pa.StructuredTetrahedralGrid([4, 4, 4], [1, 1, 1]))
    P.operator_roundtrip(lib, g, kinds=("dir", "neu"), hetero=1e6)
    P.operator_roundtrip(lib, g, kinds=("dir", "rob"), hetero=1e-6)


def test_mid_size_vs_oracle(lib):
    """~25 k tetrahedra, every matrix against the oracle (a few seconds of oracle time)."""
    g = pa.perturb_interior_nodes(_geo(pa.StructuredTetrahedralGrid([16, 16, 16], [1, 1, 1])), 0.015)
    P.operator_roundtrip(lib, g, kinds=("dir", "dir", "neu"))


def test_deterministic_bitwise(lib):
    g = pa.perturb_interior_nodes(_geo(pa.StructuredTetrahedralGrid([6, 6, 6], [1, 1, 1])), 0.03)
    d1, data1 = P.operator_roundtrip(lib, g, seed=3)
    d2, data2 = P.operator_roundtrip(lib, g, seed=3)
    for k in ("flux", "bound_flux", "vector_source", "bound_pressure_cell"):
        a = data1[pa.DISCRETIZATION_MATRICES]["flow"][k]
        b = data2[pa.DISCRETIZATION_MATRICES]["flow"][k]
        assert np.array_equal(a.data, b.data), k


def test_config_c2_scale_properties(lib):
    """BASELINE config 2 (196 608 tetrahedra): exact linear field, zero flux for constant
    pressure — size-independent properties; the oracle is too slow at this size."""
    g = _geo(pa.StructuredTetrahedralGrid([32, 32, 32], [1, 1, 1]))
    info = P.linear_field_exact(lib, g, tol=1e-10)
    assert info["iterations"] > 0


def test_tutorial_sum_and_config_c1(lib):
    from tests.test_emulation_parity import test_tutorial_sum_and_config_c1 as body
    body(lib)


def test_error_behaviour(lib):
    from tests.test_emulation_parity import test_error_behaviour_matches_reference as body
    body(lib)
