"""CPU suite for the MPSA path: host-emulation build of the kernels vs oracle / reference."""
import numpy as np
import pytest

import porepy_amd as pa
from tests import _parity as P
from tests._golden import mpsa_case_names, mpsa_contrast_case_names


@pytest.fixture(scope="module")
def lib():
    return P.emulation_library()


def _geo(g):
    g.compute_geometry()
    return g


@pytest.mark.parametrize("name", mpsa_case_names())
def test_mpsa_golden_case(lib, name):
    P.check_mpsa_golden_case(lib, name)


@pytest.mark.parametrize("key", ["cart_homogeneous", "cart_heterogeneous",
                                 "simplex_homogeneous", "simplex_heterogeneous"])
def test_mpsa_reference_known_answers(lib, key):
    P.check_mpsa_known_answer(lib, key)


@pytest.mark.parametrize("make", [
    lambda: pa.perturb_interior_nodes(_geo(pa.StructuredTriangleGrid([4, 4], [1, 1])), 0.05),
    lambda: pa.perturb_interior_nodes(_geo(pa.StructuredTetrahedralGrid([2, 2, 3], [1, 1, 1])), 0.05),
])
def test_mpsa_uniaxial_solution_is_exact(lib, make):
    # (simplex grids: with eta = 0 on Cartesian grids two rollers meeting in a corner give a
    #  singular local system in the reference as well)
    g = make()
    g.compute_geometry()
    P.mpsa_uniaxial_exact(lib, g)


@pytest.mark.parametrize("make,mode", [
    (lambda: pa.CartGrid([4, 3], [1, 1]), "roller"),
    (lambda: pa.CartGrid([3, 3, 2], [1, 1, 1]), "clamped_bottom"),
    (lambda: pa.perturb_interior_nodes(_geo(pa.StructuredTriangleGrid([4, 3], [1, 1])), 0.05), "roller"),
    (lambda: pa.perturb_interior_nodes(_geo(pa.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1])), 0.05), "roller"),
])
def test_mpsa_operator_api_vs_oracle(lib, make, mode):
    g = make()
    g.compute_geometry()
    P.mpsa_operator_roundtrip(lib, g, mode=mode)


def test_mpsa_rejects_what_it_does_not_cover(lib):
    g = _geo(pa.CartGrid([3, 3], [1, 1]))
    C = pa.FourthOrderTensor(np.ones(9), np.ones(9))
    bc = pa.BoundaryCondition(g)  # scalar bc -> same AttributeError as the reference
    data = pa.initialize_data({}, "mechanics", {"fourth_order_tensor": C, "bc": bc})
    with pytest.raises(AttributeError):
        pa.Mpsa("mechanics", library=lib).discretize(g, data)


@pytest.mark.parametrize("name", ["mpsapartial_tri2d_4x4", "mpsapartial_tet3d_2x2x2"])
def test_partial_discretization_and_update(lib, name):
    P.check_mpsa_partial_case(lib, name)


def test_amg_block_preconditioner_for_mechanics(lib):
    """bs = nd unknowns per cell: cells are aggregated, components kept apart."""
    g = pa.perturb_interior_nodes(_geo(pa.StructuredTetrahedralGrid([5, 5, 5], [1, 1, 1])), 0.03)
    info = P.mpsa_uniaxial_exact(lib, g, tol=1e-9, precond="amg")
    base = P.mpsa_uniaxial_exact(lib, g, tol=1e-9)
    assert info["iterations"] * 2 < base["iterations"], (info, base)


@pytest.mark.parametrize("name", ["biot_tri2d_3x3_mixed", "biot_cart2d_3x2_dir", "biot_tet_2x2x2_mixed",
                                  "biot_etasub_tri2d_3x3", "biot_etasub_tet_2x2x2"])
def test_biot_coupling_terms(lib, name):
    P.check_biot_case(lib, name)


def test_rotated_boundary_basis_gives_the_same_solution(lib):
    """Boundary data given in a face-wise rotated basis, u' = B u: same displacement field as the
    Cartesian statement of the same problem (tests/numerics/fv/test_mpsa.py:735-860)."""
    g = pa.perturb_interior_nodes(_geo(pa.StructuredTriangleGrid([4, 4], [1, 1])), 0.03)
    nc, nf = g.num_cells, g.num_faces
    rng = np.random.default_rng(8)
    C = pa.FourthOrderTensor(1 + rng.random(nc), 1 + rng.random(nc))
    bf = g.get_all_boundary_faces()
    gval = np.zeros((2, nf))
    gval[:, bf] = rng.random((2, bf.size)) - 0.5
    sols = []
    for rotated in (False, True):
        bc = pa.BoundaryConditionVectorial(g, bf, ["dir"] * bf.size)
        vals = gval.copy()
        if rotated:
            th = rng.random(nf) * 2 * np.pi
            bc.basis = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
            vals = np.einsum("kcf,cf->kf", bc.basis, gval)
        data = pa.initialize_data({}, "mech", {"fourth_order_tensor": C, "bc": bc, "bc_values": vals.ravel("F")})
        d = pa.Mpsa("mech", library=lib)
        d.discretize(g, data)
        u, info = d.solve(g, data, rtol=1e-13)
        sols.append(u)
    assert np.linalg.norm(sols[0] - sols[1]) <= 1e-9 * np.linalg.norm(sols[0])


@pytest.mark.parametrize("name", ["biot_tri2d_3x3_mixed", "biot_cart2d_3x2_dir", "biot_tet_2x2x2_mixed"])
def test_biot_partial_discretization_and_update(lib, name):
    P.check_biot_partial_case(lib, name)


def test_patch_parity_machinery_small(lib):
    out = P.full_size_patch_parity_mpsa(lib, 4, seeds=(0, None))
    assert out["rows_checked"] > 20


def test_biot_update_without_coupling_terms_on_the_handle_rediscretizes_fully(lib):
    """An update_discretization call when the handle only holds an MPSA discretization (empty
    scalar_vector_mappings came first): every row must come back as the full discretization's, none as zero."""
    from tests._golden import BiotCase

    c = BiotCase("biot_tri2d_3x3_mixed")
    g = pa.grid_from_raw(c.grid)

    def make(alphas):
        bc = pa.BoundaryConditionVectorial(g)
        bc.is_dir, bc.is_neu, bc.is_rob = c.bc["is_dir"].copy(), c.bc["is_neu"].copy(), c.bc["is_rob"].copy()
        bc.robin_weight = c.bc["robin_weight"]
        C = type("C", (), {"values": c.stiffness})()
        maps = {k: type("A", (), {"values": v})() for k, v in alphas.items()}
        return pa.initialize_data({}, "mechanics", {"fourth_order_tensor": C, "bc": bc, "scalar_vector_mappings": maps})

    d = pa.Biot("mechanics", library=lib)
    data = make({})
    d.discretize(g, data)  # MPSA only: no coupling terms on the handle
    data[pa.PARAMETERS]["mechanics"]["scalar_vector_mappings"] = make(c.alphas)[pa.PARAMETERS]["mechanics"]["scalar_vector_mappings"]
    data["update_discretization"] = {"modified_cells": np.array([1])}
    d.update_discretization(g, data)
    full = make(c.alphas)
    pa.Biot("mechanics", library=lib).discretize(g, full)
    md, mf = data[pa.DISCRETIZATION_MATRICES]["mechanics"], full[pa.DISCRETIZATION_MATRICES]["mechanics"]
    for k in ("stress", "bound_stress"):
        assert P.rel_max_err(md[k], mf[k]) < 1e-12
    for k in ("scalar_gradient", "displacement_divergence", "mpsa_consistency"):
        for key in c.alphas:
            assert P.rel_max_err(md[k][key], mf[k][key]) < 1e-12, (k, key)


def test_mpsa_patch_parity_machinery_small(lib):
    """The all-matrices patch test of the GPU suite on a small grid (host emulation)."""
    out = P.mpsa_patch_parity_all_matrices(lib, 4, n_random=1, reference=True)
    assert out["patches"] == 15 and out["rows_checked"] > 100
    if out["reference_patches"]:  # (the reference importable: the build container)
        assert out["reference_patches"] == 15 and max(out["worst_rel_err_vs_reference"].values()) < 1e-10


@pytest.mark.parametrize("name", ["mpsasub_cart2d_4x3", "mpsasub_tri2d_3x3_rob", "mpsasub_tet3d_2x2x2",
                                  "mpsasub_tri2d_3x3_basis_rob", "mpsasub_tet3d_2x2x2_basis",
                                  "mpsasub_tri2d_3x3_hfeta_basis", "mpsasub_cart3d_3x2x2_hfeta"])
@pytest.mark.parametrize("scramble", [False, True])
def test_boundary_conditions_per_subface(lib, name, scramble):
    P.check_mpsa_subface_case(lib, name, scramble)


@pytest.mark.parametrize("dim", [2, 3])
def test_partition_arguments_discretize_in_pieces(lib, dim):
    P.mpsa_pieces_case(lib, dim)


def test_interaction_region_larger_than_lds(lib):
    P.mpsa_large_interaction_region(lib)


@pytest.mark.parametrize("name", ["biot_tri2d_3x3_mixed", "biot_tet_2x2x2_mixed"])
def test_biot_partition_arguments_discretize_in_pieces(lib, name):
    P.biot_pieces_case(lib, name)


def test_all_four_matrices_and_the_displacement_field_on_a_whole_grid_against_the_reference(lib):
    """Whole-grid VALUE datum (oracle/gen_golden_mpsa_whole_grid.py): the reference's pp.Mpsa was run on every cell of a
    10 368-cell perturbed tetrahedral box of the configs[3] family (heterogeneous Lame parameters, rollers, traction with a
    shear component) with its own assemble_matrix_rhs and a scipy solve; the kernels (host-emulation build) reproduce the
    block digests of all four matrices and the displacement field."""
    out = P.mpsa_whole_grid_check(lib, 12)
    for k in P.MPSA_KEYS:
        assert max(out[k]) < 1e-12, (k, out[k])
    assert out["u_norm_rel_diff"] < 1e-10 and out["u_block_squares_worst_rel_diff"] < 1e-9, out
    # ... and the fine datum (oracle/gen_golden_mpsa_fine.py): sum |a| and max |a| of every block of 256 rows of the
    # reference's four matrices (observed: 1.9e-15 / 4.5e-15 over 928 blocks)
    assert out["fine"]["blocks_of_256_rows"] > 900
    assert out["fine"]["worst_rel_diff_of_block_sums"] < 1e-13 and out["fine"]["worst_rel_diff_of_block_maxima"] < 1e-12, out["fine"]


def test_biot_coupling_terms_on_a_whole_grid_against_the_reference(lib):
    """Whole-grid VALUE datum for the Biot terms (oracle/gen_golden_biot_whole_grid.py): pp.Biot run on every cell of a
    6 000-cell perturbed tetrahedral box (heterogeneous Lame parameters, anisotropic heterogeneous coupling tensor); the
    kernels (host-emulation build) reproduce the block digests of the five coupling matrices and of stress / bound_stress."""
    out = P.biot_whole_grid_check(lib, 10)
    for k, v in out.items():
        if isinstance(v, list):
            assert max(v) < 1e-12, (k, v)


@pytest.mark.parametrize("name", mpsa_contrast_case_names())
def test_stiffness_contrasts_of_1e8_to_1e12_between_neighbouring_cells(lib, name):
    assert P.check_mpsa_contrast_case(lib, name) < 1e-10


def test_the_fp64_body_alone_misses_the_contrast_fixtures(lib):
    assert P.mpsa_contrast_fp64_body_misses(lib) > 1e-9


def test_mechanics_system_replays_its_positions_under_kept_patterns(lib):
    assert P.mpsa_assemble_positions_replayed(lib)


def test_singular_corner_region_is_reported_not_faulted(lib):
    assert P.mpsa_singular_corner_is_an_error_not_a_fault(lib)
