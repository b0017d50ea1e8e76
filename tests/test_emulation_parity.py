"""CPU suite: the kernel sources compiled as a sequential host emulation (oracle/_build,
test infrastructure) driven through the same C ABI and checked against the oracle, the
reference's golden matrices and its known-answer vectors.  Covers the host logic too."""
import numpy as np
import pytest

import porepy_amd as pa
from tests import _parity as P
from tests._golden import periodic_case_names, case_names


@pytest.fixture(scope="module")
def lib():
    return P.emulation_library()


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
    lambda: pa.CartGrid([6, 5], [1, 1]),
    lambda: pa.CartGrid([4, 3, 3], [1, 1, 1]),
    lambda: pa.perturb_interior_nodes(_geo(pa.CartGrid([5, 5], [1, 1])), 0.05),
    lambda: pa.perturb_interior_nodes(_geo(pa.StructuredTriangleGrid([5, 4], [1, 1])), 0.05),
    lambda: pa.perturb_interior_nodes(_geo(pa.StructuredTetrahedralGrid([3, 3, 2], [1, 1, 1])), 0.06),
])
def test_operator_api_on_own_grids(lib, make):
    g = make()
    g.compute_geometry()
    P.operator_roundtrip(lib, g)


def _geo(g):
    g.compute_geometry()
    return g


def test_heterogeneous_1e6(lib):
    g = _geo(pa.StructuredTetrahedralGrid([3, 3, 3], [1, 1, 1]))
    P.operator_roundtrip(lib, g, kinds=("dir", "neu"), hetero=1e6)
    P.operator_roundtrip(lib, g, kinds=("dir", "rob"), hetero=1e-6)


@pytest.mark.parametrize("make", [
    lambda: pa.CartGrid([7, 6], [1, 1]),
    lambda: pa.CartGrid([4, 4, 4], [1, 1, 1]),
    lambda: pa.perturb_interior_nodes(_geo(pa.StructuredTetrahedralGrid([4, 4, 4], [1, 1, 1])), 0.05),
])
def test_linear_field_exact(lib, make):
    g = make()
    g.compute_geometry()
    P.linear_field_exact(lib, g)


def test_cartesian_laplacian_is_symmetric_and_cg_works(lib):
    g = _geo(pa.CartGrid([8, 8], [1, 1]))
    K = pa.SecondOrderTensor(np.ones(g.num_cells))
    bf = g.get_all_boundary_faces()
    bc = pa.BoundaryCondition(g, bf, ["dir"] * bf.size)
    data = pa.initialize_data({}, "flow", {"second_order_tensor": K, "bc": bc,
                                           "bc_values": np.zeros(g.num_faces)})
    d = pa.Mpfa("flow", library=lib)
    d.discretize(g, data)
    A, b = d.assemble_matrix_rhs(g, data)
    assert abs(A - A.T).max() < 1e-13
    x, info = d.solve(g, data, source=g.cell_volumes, method="cg", rtol=1e-12)
    assert info["converged"]
    assert np.linalg.norm(A @ x - g.cell_volumes) < 1e-11 * np.linalg.norm(g.cell_volumes)


def test_tutorial_sum_and_config_c1(lib):
    """tutorials/flux_discretizations.ipynb cell 30 and BASELINE config 1 of the reference."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "scalar_known_answers.npz"))
    g = _geo(pa.CartGrid([20, 20], [1, 1]))
    K = pa.SecondOrderTensor(np.ones(g.num_cells))
    bf = g.get_all_boundary_faces()
    bc = pa.BoundaryCondition(g, bf, ["dir"] * bf.size)
    data = pa.initialize_data({}, "flow", {"second_order_tensor": K, "bc": bc,
                                           "bc_values": np.zeros(g.num_faces)})
    d = pa.Mpfa("flow", library=lib)
    d.discretize(g, data)
    p, _ = d.solve(g, data, source=g.cell_volumes, rtol=1e-13)
    assert np.isclose(p.sum(), float(z["tutorial_sum_documented"]))
    assert abs(p.sum() - float(z["tutorial_sum_reference_today"])) < 1e-10 * p.sum()
    g = _geo(pa.CartGrid([50, 50], [1, 1]))
    K = pa.SecondOrderTensor(np.ones(g.num_cells))
    bf = g.get_all_boundary_faces()
    west = bf[g.face_centers[0, bf] < 1e-10]
    east = bf[g.face_centers[0, bf] > 1 - 1e-10]
    bc = pa.BoundaryCondition(g, np.r_[west, east], ["dir"] * (west.size + east.size))
    bv = np.zeros(g.num_faces)
    bv[west], bv[east] = 5.0, 2.0
    data = pa.initialize_data({}, "flow", {"second_order_tensor": K, "bc": bc, "bc_values": bv})
    d = pa.Mpfa("flow", library=lib)
    d.discretize(g, data)
    p, info = d.solve(g, data, rtol=1e-13)
    assert abs(p.sum() - float(z["c1_sum_reference_today"])) < 1e-9 * p.sum()
    assert np.allclose([p.min(), p.max()], z["c1_min_max"])


def test_error_behaviour_matches_reference(lib):
    # singular local system -> ValueError with the reference's message
    g = _geo(pa.CartGrid([3, 3], [1, 1]))
    K = pa.SecondOrderTensor(np.zeros(g.num_cells))
    bf = g.get_all_boundary_faces()
    bc = pa.BoundaryCondition(g, bf, ["dir"] * bf.size)
    data = pa.initialize_data({}, "flow", {"second_order_tensor": K, "bc": bc})
    with pytest.raises(ValueError, match="Error in inversion of local linear systems"):
        pa.Mpfa("flow", library=lib).discretize(g, data)
    # bad call order
    ctx = pa.Context(0, lib)
    with pytest.raises(pa.PorefvError):
        ctx.discretize()


def test_rediscretize_with_new_parameters_reuses_topology(lib):
    g = _geo(pa.StructuredTriangleGrid([4, 4], [1, 1]))
    d, data = P.operator_roundtrip(lib, g, seed=1)
    st1 = d.context(g).stats()
    data[pa.PARAMETERS]["flow"]["second_order_tensor"] = pa.SecondOrderTensor(2 * np.ones(g.num_cells))
    f_old = data[pa.DISCRETIZATION_MATRICES]["flow"]["flux"].copy()
    d.discretize(g, data)
    f_new = data[pa.DISCRETIZATION_MATRICES]["flow"]["flux"]
    assert abs(f_new - f_old).max() > 1e-3
    assert d.context(g).stats()["num_sub_half_faces"] == st1["num_sub_half_faces"]


def test_vector_source_matrices_addressed_through_the_flux_pattern(lib):
    assert P.implicit_vector_source_pattern(lib)


def test_rebuilt_topology_keeps_the_patterns_it_proves_unchanged(lib):
    assert P.symbolic_reuse_on_rebuilt_topology(lib)


def test_ready_run_major_face_order_leaves_the_same_matrices(lib):
    # (the host build has no second stream: what is pinned here is the face order the pipeline's topology produces)
    assert P.node_face_pipeline_leaves_the_same_bits(lib, 8, device=False)


@pytest.mark.parametrize("name", ["partial_cart2d_5x5", "partial_tet3d_3x3x3"])
def test_partial_discretization_and_update(lib, name):
    P.check_partial_case(lib, name)


def test_partial_discretization_one_cell_at_a_time(lib):
    P.partial_one_cell_at_a_time(lib)


@pytest.mark.parametrize("name", ["tilted_cart2d_4x3", "tilted_tri2d_4x4", "tilted_flat_tri2d_3x3", "tilted_vdim2_cart2d_4x3", "tilted_vdim2_tri2d_4x4",
                                  # subdomains of a mixed-dimensional grid (oracle/gen_golden_md.py): 3-D matrix with
                                  # fracture faces, 2-D fracture grids in the planes x = 0.5 / y = 0.5
                                  "tilted_md_box_matrix3d", "tilted_md_box_fracture0", "tilted_md_box_fracture1"])
def test_2d_grid_embedded_in_3d(lib, name):
    P.check_tilted_case(lib, name)


@pytest.mark.parametrize("restart", [0, 7])
def test_gmres(lib, restart):
    g = pa.perturb_interior_nodes(_geo(pa.StructuredTetrahedralGrid([4, 4, 4], [1, 1, 1])), 0.04)
    P.gmres_matches_direct(lib, g, restart=restart)
    g = _geo(pa.CartGrid([9, 7], [1, 1]))
    P.gmres_matches_direct(lib, g, restart=restart)


def test_solve_in_place_on_morton_numbered_grid(lib):
    its = P.morton_numbered_grid_solve(lib)
    assert its["morton", "amg"] * 3 < its["morton", "jacobi"]


def test_solver_for_assembled_csr_systems(lib):
    """pfv_set_system: any CSR system with a non-zero diagonal (rows not sorted, no grid)."""
    import scipy.sparse as sps
    import scipy.sparse.linalg as spla

    rng = np.random.default_rng(3)
    n = 400
    A = sps.random(n, n, density=0.02, random_state=5, format="csr") + sps.diags(4 + rng.random(n))
    A = sps.csr_matrix(A)
    perm = np.concatenate([rng.permutation(np.arange(A.indptr[i], A.indptr[i + 1])) for i in range(n)])
    A = sps.csr_matrix((A.data[perm], A.indices[perm], A.indptr), shape=A.shape)  # unsorted columns
    b = rng.random(n)
    xo = spla.spsolve(A.tocsc(), b)
    for method in ("bicgstab", "gmres"):
        x, info = pa.solve_csr(A, b, method=method, rtol=1e-13, library=lib)
        assert info["converged"]
        assert np.linalg.norm(x - xo) <= 1e-10 * np.linalg.norm(xo), method
    Z = A.tolil()
    Z[7, 7] = 0.0
    with pytest.raises(pa.PorefvError) as e:
        pa.solve_csr(Z.tocsr(), b, library=lib)
    assert e.value.status == 5 and "row 7" in e.value.message


@pytest.mark.parametrize("name", ["tpfa_line_8", "tpfa_line_6_in_3d_via_mpfa", "tpfa_cart2d_4x3", "tpfa_tet3d_2x2x2"])
def test_tpfa_and_1d_delegation(lib, name):
    P.check_tpfa_case(lib, name)


def test_zero_dimensional_grid(lib):
    P.check_zero_dimensional_grid(lib)


def test_grids_discretized_as_disjoint_unions_leave_the_bits_of_the_single_grid_path(lib):
    P.batch_matches_single(lib)


def test_batch_hands_special_inputs_to_the_single_grid_path(lib):
    P.batch_hands_special_inputs_to_the_single_grid_path(lib)


def test_amg_filter_keeps_its_row_layout_only_after_a_setup_that_reproduced_it(lib):
    P.amg_filter_layout_states(lib)


@pytest.mark.parametrize("env", [{"PFV_AMG_FUSE_ROWS": "0", "PFV_AMG_GAMMA": "2"},
                                 {"PFV_AMG_FUSE_ROWS": "0", "PFV_AMG_GAMMA": "1"},
                                 {"PFV_AMG_FUSE_ROWS": "0", "PFV_AMG_GAMMA": "2", "PFV_AMG_GAMMA_LEVELS": "2"},
                                 {"PFV_AMG_FUSE_ROWS": "1000", "PFV_AMG_GAMMA": "2", "PFV_AMG_GAMMA_LEVELS": "2"}])
def test_amg_fused_cycle_is_the_same_operator(lib, env):
    g = pa.StructuredTetrahedralGrid([18, 18, 18], [1, 1, 1])
    g.compute_geometry()
    g = pa.perturb_interior_nodes(g, 0.015)
    out = P.amg_fused_cycle_is_the_same_operator(lib, g, env=env)
    assert out["levels"] >= 4 and out["iterations"][0] == out["iterations"][1], out


def test_amg_preconditioner(lib):
    g = pa.perturb_interior_nodes(_geo(pa.StructuredTetrahedralGrid([10, 10, 10], [1, 1, 1])), 0.02)
    out, jac, st = P.amg_preconditioner(lib, g)
    # strongly heterogeneous Cartesian grid: matching runs out of strong neighbours early
    g = _geo(pa.CartGrid([14, 14, 14], [1, 1, 1]))
    P.amg_preconditioner(lib, g, hetero_sigma=2.0)
    g = _geo(pa.CartGrid([40, 30], [1, 1]))
    P.amg_preconditioner(lib, g)


def test_amg_on_assembled_csr_and_symmetric_cg(lib):
    import scipy.sparse as sps
    import scipy.sparse.linalg as spla

    n = 60
    T = sps.diags([-1, 2, -1], [-1, 0, 1], shape=(n, n))
    A = (sps.kron(sps.identity(n), T) + sps.kron(T, sps.identity(n))).tocsr()
    b = np.ones(n * n)
    xo = spla.spsolve(A.tocsc(), b)
    xj, ij = pa.solve_csr(A, b, method="cg", rtol=1e-12, library=lib)
    xa, ia = pa.solve_csr(A, b, method="cg", rtol=1e-12, library=lib, precond="amg")
    assert np.linalg.norm(xa - xo) <= 1e-10 * np.linalg.norm(xo)
    assert ia["iterations"] * 2 < ij["iterations"]
    # the same system in SI-like units (entries ~1e-15): the hierarchy -- incl. the dense inverse of the coarsest
    # level, whose pivot test is an absolute one on row-scaled systems -- must not depend on the units
    xs, isc = pa.solve_csr(A * 1e-15, b * 1e-15, method="cg", rtol=1e-12, library=lib, precond="amg")
    assert isc["iterations"] == ia["iterations"], (isc["iterations"], ia["iterations"])
    assert np.linalg.norm(xs - xo) <= 1e-10 * np.linalg.norm(xo)


@pytest.mark.parametrize("name", ["subface_cart2d_4x3", "subface_tet3d_2x2x2"])
@pytest.mark.parametrize("scramble", [False, True])
def test_boundary_conditions_per_subface(lib, name, scramble):
    P.check_subface_case(lib, name, scramble)


@pytest.mark.parametrize("name", ["persub_cart2d_4x5", "persub_tri2d_4x4", "persub_tet3d_2x2x3"])
@pytest.mark.parametrize("scramble", [False, True])
def test_boundary_conditions_per_subface_on_a_grid_with_periodic_faces(lib, name, scramble):
    P.check_periodic_subface_case(lib, name, scramble)


def test_device_resident_vectors(lib):
    g = pa.perturb_interior_nodes(_geo(pa.StructuredTetrahedralGrid([4, 4, 4], [1, 1, 1])), 0.03)
    P.device_resident_vectors(lib, g)


@pytest.mark.parametrize("scheme", ["mpfa", "tpfa"])
@pytest.mark.parametrize("name", periodic_case_names())
def test_periodic_faces(lib, name, scheme):
    P.check_periodic_case(lib, name, scheme)


def test_periodic_faces_error_behaviour(lib):
    g = _geo(pa.CartGrid([3, 3], [1, 1]))
    g.set_periodic_map(np.array([[13, 12, 14], [21, 22, 23]]))  # not sorted: as _fvutils.py:103-112
    K = pa.SecondOrderTensor(np.ones(g.num_cells))
    data = pa.initialize_data({}, "flow", {"second_order_tensor": K, "bc": pa.BoundaryCondition(g)})
    with pytest.raises(NotImplementedError):
        pa.Mpfa("flow", library=lib).discretize(g, data)
    g = _geo(pa.CartGrid([3, 3], [1, 1]))
    g.set_periodic_map(np.array([[21, 22, 23], [12, 13, 14]]))  # left faces on the higher cells
    data = pa.initialize_data({}, "flow", {"second_order_tensor": K, "bc": pa.BoundaryCondition(g)})
    with pytest.raises(NotImplementedError):
        pa.Mpfa("flow", library=lib).discretize(g, data)
    bcv = pa.BoundaryConditionVectorial(g)
    C = pa.FourthOrderTensor(np.ones(g.num_cells), np.ones(g.num_cells))
    data = pa.initialize_data({}, "mech", {"fourth_order_tensor": C, "bc": bcv})
    with pytest.raises(NotImplementedError):  # mpsa.py:661-664
        pa.Mpsa("mech", library=lib).discretize(g, data)


def test_patch_parity_machinery_small(lib):
    """The full-size GPU test's machinery on a small grid (host emulation)."""
    out = P.full_size_patch_parity(lib, 6, seeds=(0, None))
    assert out["rows_checked"] > 20


def test_bench_grid_patch_parity_machinery_small(lib):
    """The machinery of the full-size GPU tests (all six matrices + A on patches of the grid bench.py times,
    and of the configs[1] lattice) on small grids (host emulation)."""
    out = P.bench_grid_patch_parity(lib, 5, n_random=2)
    assert out["patches"] == 16 and out["rows_checked"] > 200
    assert out["true_rel_residual"] < 1e-12
    # the same rows against the reference itself (pp.Mpfa on the same patches), where it is importable
    if out["reference_patches"]:
        assert out["reference_patches"] == 16 and out["patterns_bit_exact_vs_reference"] == 32
        assert max(out["worst_rel_err_vs_reference"].values()) < 1e-10
    out = P.config_c2_patch_parity(lib, 4, n_random=2)
    assert out["max_abs_error_vs_exact_linear_field"] < 1e-10
    if out["reference_patches"]:
        assert max(out["worst_rel_err_vs_reference"].values()) < 1e-10


@pytest.mark.parametrize("with_vs", [True, False])
def test_ad_flux_system_vs_oracle(lib, with_vs):
    """Residual + Jacobian of the flow equation with K = K(p), assembled on the device (N4)."""
    out = P.check_ad_flux_system(lib, 3, with_vs=with_vs)
    assert out["nnz_J"] > 0


def test_amg_robustness_sweep_small(lib):
    """The sweep of the GPU suite at a quarter of the size (host emulation)."""
    out = P.amg_robustness_sweep(lib, scale=0.25)
    for k, (n, its, res) in out.items():
        assert res < 1.05e-10 and its <= 40, (k, n, its, res)


def test_interaction_region_with_more_than_64_subfaces(lib):
    P.mpfa_large_interaction_region(lib)


def test_sliver_grids_take_the_iterative_refinement_path(lib):
    P.sliver_refinement(lib)
