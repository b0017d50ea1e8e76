"""Drop-in checks against the reference package ITSELF: unmodified PorePy models run with ``pp.Mpfa`` /
``pp.Mpsa`` / ``pp.Biot`` rebound to the operators of this package.

Two variants of every test:
* ``emulation`` (``-m "not gpu"``, build container: the live reference tree, no GPU) binds the host-emulation
  build of the kernel sources;
* ``product`` (``-m gpu``, the GPU box) binds **libporefv_hip.so**, with the reference imported from the
  byte-compiled archive ``oracle/_ref/porepy_ref.zip`` that ``oracle/make_ref.py`` builds from the
  reference tree where it lies (``__graft_entry__.build()`` runs the recipe; the archive is git-ignored
  and travels with the gpurun snapshot).  This is the call chain ``pp.ad.MpfaAd`` ->
  ``pp.Mpfa`` rebind (/root/reference/src/porepy/numerics/ad/discretizations.py:195) -> the HIP library.
Skipped where no reference is importable."""
import json
import os
import subprocess
import sys

import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = [pytest.param("emulation", id="emulation"),
            pytest.param("product", id="product", marks=pytest.mark.gpu)]


def run_script(name: str, variant: str, timeout: int = 900, extra_env=None, args=()):
    env = oracle.ref_env(extra_last=[ROOT], prefer_archive=(variant == "product"))
    if env is None:
        pytest.skip("reference PorePy not present (neither /root/reference nor oracle/_ref/porepy_ref.zip)")
    env["PFV_DROPIN_LIBRARY"] = variant
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", name), *args], env=env, cwd="/tmp",
                       capture_output=True, text=True, timeout=timeout)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stderr[-3000:]
    out = json.loads(line[-1][7:])
    want = "libporefv_hip.so" if variant == "product" else "libporefv_emul.so"
    assert os.path.basename(out["library"]) == want, out["library"]
    return out


@pytest.mark.parametrize("variant", VARIANTS)
def test_single_phase_flow_model_with_rebound_mpfa(variant):
    out = run_script("_dropin_script.py", variant, 600)
    assert out["cells"] == 2500
    assert out["calls_into_device_path"] >= 1
    assert out["p_rel_err"] < 1e-10
    assert out["A_rel_err"] < 1e-10  # (the final residual vector is round-off in both runs)
    assert out["p_rel_err_hip_solver"] < 1e-10 and out["hip_solver_iterations"] > 0
    assert abs(out["p_sum_ref"] - 8750.0) < 1e-6  # config C1 of the reference
    # the whole Newton step in HBM under the unmodified model: matrices kept on the device (as_porepy_discretization(lazy=True)),
    # operator trees walked with device Jacobians and device matrix leaves (DeviceAssembly), the device Jacobian solved by
    # the device Krylov solver; the flux matrix is never fetched to the host
    d = out["all_on_device"]
    assert "error" not in d, d
    assert d["jacobian_on_device"] and d["solved_from_device_jacobian"] and d["iterations"] > 0
    assert d["flux_proxy"] == "LazyCsr" and d["flux_fetched_to_host"] is False
    assert d["p_rel_err"] < 1e-10 and d["A_rel_err"] < 1e-12


@pytest.mark.parametrize("variant", VARIANTS)
def test_momentum_balance_and_poromechanics_models_with_rebound_mpsa_biot(variant):
    """pp.Mpsa, pp.Biot and pp.Mpfa rebound under the reference's own MomentumBalance and
    Poromechanics models: displacement / pressure and the Jacobian reproduce the untouched runs."""
    out = run_script("_dropin_mech_script.py", variant)
    assert out["calls"]["mpsa"] >= 1 and out["calls"]["biot"] >= 1
    assert out["mech_dofs"] == 72 and out["poro_dofs"] == 108
    assert out["mech_x_rel_err"] < 1e-10 and out["mech_A_rel_err"] < 1e-10
    assert out["poro_x_rel_err"] < 1e-10 and out["poro_A_rel_err"] < 1e-10


@pytest.mark.parametrize("variant", VARIANTS)
def test_differentiable_tpfa_transmissibilities_in_the_reference_model(variant):
    """The reference's unit-test model of its differentiable TPFA flux (two cells, pressure-dependent
    full-tensor permeability): ``AdTpfaFlux.__transmissibility_matrix`` through its operator tree and
    forward AD against pfv_tpfa_transmissibility_ad chained with the Jacobian of k_c."""
    out = run_script("_dropin_adtpfa_script.py", variant, 600)
    for base in ("tpfa", "mpfa"):
        o = out[base]
        assert o["faces"] == 7 and o["dofs"] == 2 and o["jac_nnz_ref"] > 0
        assert o["t_rel_err"] < 1e-12 and o["jac_rel_err"] < 1e-12
        # the mixin of porepy_amd.as_porepy_ad_tpfa_flux() in front of the reference's classes: flux and
        # pressure trace of the whole model, values and Jacobians
        m = out[base + "_model"]
        assert m["device_path_calls"] >= 1 and m["flux_jac_nnz"] > 0
        assert max(m["flux_rel_err"], m["flux_jac_rel_err"], m["trace_rel_err"], m["trace_jac_rel_err"]) < 1e-12


@pytest.mark.parametrize("variant", VARIANTS)
def test_mixed_dimensional_model_with_rebound_mpfa_and_hip_solver(variant):
    """The structured stand-in for BASELINE configs[4]: the reference's SinglePhaseFlow on a 3-D box with two
    intersecting fractures (2-D subdomains, 1-D intersection, mortar grids), pp.Mpfa rebound and the coupled
    Jacobian solved by the HIP Krylov solver: all unknowns (matrix, fractures, intersection, mortar fluxes)
    and the Jacobian reproduce the untouched run."""
    out = run_script("_dropin_md_script.py", variant)
    assert out["dims"] == [3, 2, 1] and out["subdomains"] == 4 and out["interfaces"] == 4 and out["mortar_cells"] > 0
    assert out["device_calls_by_dim"]["3"] >= 1 and out["device_calls_by_dim"]["2"] >= 2
    assert out["x_rel_err"] < 1e-10 and out["A_rel_err"] < 1e-10
    assert out["x_rel_err_hip_solver"] < 1e-10 and out["hip_solver_iterations"] > 0
    # the whole Newton step on the device for the mixed-dimensional model too (DeviceAssembly + lazy matrices + device leaves)
    d = out["all_on_device"]
    assert "error" not in d, d
    assert d["jacobian_on_device"] and d["solved_from_device_jacobian"] and d["iterations"] > 0
    assert d["x_rel_err"] < 1e-10 and d["A_rel_err"] < 1e-12


@pytest.mark.parametrize("variant", VARIANTS)
def test_thermo_hydro_mixed_dimensional_model_with_rebound_mpfa(variant):
    """BASELINE configs[4] as north_star states it -- thermo-hydro coupling on the mixed-dimensional grid: the
    reference's MassAndEnergyBalance (models/mass_and_energy_balance.py:83) on the 2-fracture stand-in, Darcy AND
    Fourier flux discretized through the rebound ``pp.Mpfa`` (models/constitutive_laws.py:1078-1088, 2425-2437) on
    every subdomain of dimension >= 2; two implicit time steps of Newton iterations reproduce the untouched run."""
    out = run_script("_dropin_thermal_script.py", variant)
    assert out["variables"] == ["interface_darcy_flux", "interface_enthalpy_flux", "interface_fourier_flux",
                                "pressure", "temperature"]
    assert out["dims"] == [3, 2, 1] and out["subdomains"] == 4 and out["interfaces"] == 4
    c = out["device_calls"]
    assert c["flow:3"] >= 1 and c["flow:2"] >= 2 and c["fourier_discretization:3"] >= 1 and c["fourier_discretization:2"] >= 2
    assert out["T_range"][1] - out["T_range"][0] > 1.0 and out["p_range"][1] - out["p_range"][0] > 0.5  # a non-trivial state
    assert max(out["x_rel_err"], out["T_rel_err"], out["p_rel_err"], out["A_rel_err"]) < 1e-10
    # the linear systems of the Newton iterations -- equations ordered differently from the unknowns (most diagonal
    # entries structurally zero), mortar coupling: what the reference hands to a direct solver -- solved on the
    # device: GMRES + block lower-triangular preconditioner (porepy_amd.solvers.solve_block_system)
    assert out["hip_linear_solves"] >= 4 and out["hip_rows_matched"] and out["hip_solver_blocks"] >= 10
    assert out["hip_solver_worst_true_residual"] < 1e-11
    assert out["x_rel_err_hip_solver"] < 1e-10
    # ... and with the coupled Jacobian assembled on the device (DeviceAssembly), the block solver on top of it
    d = out["all_on_device"]
    assert "error" not in d, d
    assert d["jacobian_on_device"] and d["x_rel_err"] < 1e-10 and d["max_iterations"] <= 120


@pytest.mark.parametrize("variant", VARIANTS)
def test_c5_thermo_hydro_model_on_a_52_fracture_network(variant):
    """BASELINE configs[4] at its stated network size: 52 fractures in a 3-D box (114 intersection lines, 23 points,
    385 interfaces; structured stand-in for the gmsh geometry), the reference's MassAndEnergyBalance with ``pp.Mpfa``
    rebound on all 190 subdomains and every Newton system (21 360 unknowns) solved on the device: GMRES on the system
    with the interface fluxes condensed, variable-wide AMG blocks (porepy_amd.solvers: eliminate)."""
    out = run_script("_dropin_c5_script.py", variant, 1500)
    assert (out["fractures"], out["lines"], out["points"], out["interfaces"]) == (52, 114, 23, 385)
    assert out["dofs"] == 21360
    # the subdomain loop runs through md_sharding.batched_discretization: per keyword (Darcy, Fourier) the 52 fracture
    # planes are ONE disjoint union on the device, the 3-D grid -- alone in its dimension -- takes the single-grid path
    c = out["device_calls"]
    assert c["batch_calls"] == 2 and c["device_unions"] == 2 and c["grids_in_unions"] == 104 and c["grids_alone"] == 2
    assert out["loop"]["jobs_in_batches"] == 380  # (1-D and 0-D grids included: the class hands them on as upstream)
    assert out["T_range"][1] - out["T_range"][0] > 1.0
    assert out["hip_linear_solves"] >= 3 and out["hip_solver_blocks"] == 5
    assert out["hip_solver_max_iterations"] <= 120 and out["hip_solver_worst_true_residual"] < 1e-11
    assert max(out["x_rel_err"], out["T_rel_err"], out["p_rel_err"], out["A_rel_err"]) < 1e-10


def test_c5_discretization_sharded_by_subdomain_under_gloo():
    """The reference's discretization loop (numerics/ad/ad_utils.py:281-308) dealt out to two ranks by subdomain
    (porepy_amd.md_sharding), each rank discretizing its share through the rebound ``pp.Mpfa`` and one exchange of the
    stored matrices: on every rank the model's solution and last Jacobian are BITWISE those of the serial loop, and
    the device calls are split between the ranks.  (Network of 16 fractures: the model set-up of the reference, not
    the discretization, is what takes the time here.)"""
    env = oracle.ref_env(extra_last=[ROOT])
    if env is None:
        pytest.skip("reference PorePy not present")
    env.update({"PFV_DROPIN_LIBRARY": "emulation", "C5_FRACTURES": "16", "C5_N_SIDE": "10", "C5_MAX_EXTENT": "6",
                "OMP_NUM_THREADS": "2"})
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "_dropin_c5_script.py"), "--sharded"],
                       env=env, cwd="/tmp", capture_output=True, text=True, timeout=1500)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stderr[-3000:]
    out = json.loads(line[-1][7:])
    assert out["world"] == 2 and out["fractures"] == 16
    ranks = out["ranks"]
    assert all(x["same"] for x in ranks)
    total = ranks[0]["device_calls_serial"]
    assert total > 40 and sum(x["device_calls_here"] for x in ranks) == total
    assert all(0 < x["device_calls_here"] < total for x in ranks)
    first = ranks[0]["stats"]["plans"][0]
    # the 3-D grid's two interaction-region jobs (Darcy, Fourier) bound the speed-up of this loop: one on each rank
    assert 1.5 < first["bound_total_over_largest_job"] < 4.0 and first["speedup_by_cost_model"] > 1.5
    assert all(x["stats"]["matrix_bytes_sent"] > 0 for x in ranks)
    # inside a rank the fracture planes it owns went to the device as disjoint unions
    assert sum(x["device_unions"] for x in ranks) >= 2


def test_c5_matrix_grid_cut_into_cell_pieces_inside_the_sharded_loop_under_gloo():
    """the pieces composed.  At four ranks the two jobs of the 3-D matrix grid (Darcy, Fourier) would
    bound the loop at ~2.3x; ``md_sharding.plan`` cuts each into cell pieces (Morton partition + one node ring,
    ``distributed.extract_subdomain``) that are dealt out together with the fracture jobs, every rank discretizes its
    pieces through the rebound ``pp.Mpfa`` (``discretize_piece``), the rows travel in the loop's single exchange and are
    merged on every rank.  The model's solution and last Jacobian equal those of the serial loop to 1e-10 on every rank
    (rows of faces between two pieces are averaged: not bitwise), no rank repeats the whole 3-D grid, and the bound of
    the plan by the cost model exceeds 6 at eight ranks."""
    env = oracle.ref_env(extra_last=[ROOT])
    if env is None:
        pytest.skip("reference PorePy not present")
    env.update({"PFV_DROPIN_LIBRARY": "emulation", "C5_FRACTURES": "16", "C5_N_SIDE": "10", "C5_MAX_EXTENT": "6",
                "OMP_NUM_THREADS": "2", "C5_SPLIT": "1"})
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "_dropin_c5_script.py"), "--sharded"],
                       env=env, cwd="/tmp", capture_output=True, text=True, timeout=2400)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stderr[-3000:]
    out = json.loads(line[-1][7:])
    assert out["world"] == 4 and out["fractures"] == 16
    ranks = out["ranks"]
    assert all(x["x_rel_err"] < 1e-10 and x["A_rel_err"] < 1e-10 for x in ranks), [(x["x_rel_err"], x["A_rel_err"]) for x in ranks]
    first = ranks[0]["stats"]["plans"][0]
    assert first["subdomains_cut_into_pieces"] == 2 and first["piece_jobs"] >= 4, first
    assert first["bound_total_over_largest_job"] > 4.0 and first["speedup_by_cost_model"] > 3.0, first
    assert all(x["stats"]["matrix_bytes_sent"] > 0 for x in ranks)
    # ... and the Newton systems of that run were solved sharded over the same four ranks
    assert all(len(x["sharded_solves"]) >= 3 and all(q["world"] == 4 for q in x["sharded_solves"]) for x in ranks)


@pytest.mark.parametrize("variant", VARIANTS)
def test_merged_operator_parse_and_flux_products_on_the_device(variant):
    """``MergedOperator.parse`` (numerics/ad/ad_utils.py:597-663) and the matrix products of the flux
    expression formed by ``porepy_amd.DeviceCsr`` on the reference's mixed-dimensional model.  The device concatenation
    of the reference's own blocks IS its ``parse`` result (bit for bit); with the blocks of the 3-D subdomain resident
    on the device (never copied to the host) the merged matrices, the Jacobian blocks Div Flux and
    Div BoundFlux P_mortar and the flux of a random state agree with scipy on the reference's matrices to the
    parity tolerance of the discretization."""
    out = run_script("_dropin_merged_script.py", variant, 600)
    assert out["subdomains"] == 4 and out["dims"] == [3, 2, 1]
    assert out["block_diag_bit_identical_to_reference_parse"] is True
    assert all(v == "device" for v in out["where_the_blocks_were"]["dim3"].values())
    # ... and so were those of the fracture planes (2-D grids embedded in 3-D): the lift of their vector-source columns
    # into the ambient space is a product on the device
    assert all(v == "device" for v in out["where_the_blocks_were"]["dim2"].values())
    assert max(out["merged_rel_err"].values()) < 1e-10
    assert out["J_pp_rel_err"] < 1e-10 and out["J_pl_rel_err"] < 1e-10 and out["flux_rel_err"] < 1e-10
    assert out["J_pp_shape"] == [100, 100] and out["J_pl_nnz"] > 0


@pytest.mark.parametrize("variant", VARIANTS)
def test_operator_trees_of_the_reference_evaluated_with_device_jacobians(variant):
    """``EquationSystem.assemble`` (numerics/ad/equation_system.py:1579) against
    ``porepy_amd.ad.assemble_on_device`` -- the reference's own parser (numerics/ad/_ad_parser.py) and ``AdArray``
    arithmetic (numerics/ad/forward_mode.py) walking the model's operator trees from an identity whose Jacobian is a
    ``DeviceCsr``: every Jacobian product, row scaling and sum of the tree runs on the device.  The mixed-dimensional
    compressible flow model at a perturbed state: residual and Jacobian bit for bit; the thermo-hydro model (7
    equations, upwinding, density and enthalpy functions): same pattern, values to the last bit or two."""
    out = run_script("_dropin_adtree_script.py", variant, 600)
    f = out["md_flow"]
    assert f["dims"] == [3, 2, 1] and f["dofs"] == 180 and f["equations"] == 3
    assert f["bit_identical"] is True and f["rhs_identical"] is True
    assert out["newton_increment_rel_err"] < 1e-10 and out["newton_increment_iterations"] > 0
    # the discretization matrices entering the trees as DeviceCsr too (ad.device_matrix_leaves: MergedOperator.parse on the
    # device, values by device SpMV): the same Jacobian, the residual to rounding
    dl = out["md_flow_device_leaves"]
    assert dl["same_shape"] and dl["jac_rel_err"] < 1e-15 and dl["rhs_rel_err"] < 1e-13
    t = out["thermo_hydro"]
    assert "error" not in t, t
    assert t["dofs"] == 440 and t["equations"] == 7 and t["same_nonzero_pattern"] is True and t["rhs_identical"] is True
    assert t["jac_rel_err"] < 1e-15


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("mode, decades, seeds", [("", None, 6), ("special", None, 4), ("contrast", "2,6", 8)])
def test_fixed_seed_slice_of_the_differential_driver_against_the_reference(variant, mode, decades, seeds):
    """``tools/fuzz_vs_reference.py`` (random grids, tensors, conditions through PorePy itself and through the operator
    classes of this package) on a fixed slice of seeds: plain, the special legs (conditions per sub-face, partial
    discretization, tilted grids, TPFA, continuity points per sub-face), and permeability / stiffness contrasts of
    1e2 ... 1e6 between neighbouring cells -- where a verdict of the local inversions differs it must be a singular or
    near-singular INPUT by the 60-digit inversion of the reference's own systems.  No suspicious case."""
    env = oracle.ref_env(extra_last=[ROOT], prefer_archive=(variant == "product"))
    if env is None:
        pytest.skip("reference PorePy not present (neither /root/reference nor oracle/_ref/porepy_ref.zip)")
    if variant == "product":
        env["PFV_FUZZ_DEVICE"] = "1"
    if decades:
        env["PFV_FUZZ_DECADES"] = decades
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    args = [str(seeds), "424242"] + ([mode] if mode else [])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_vs_reference.py"), *args], env=env, cwd="/tmp",
                       capture_output=True, text=True, timeout=900)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert tail == "suspicious cases: 0", (r.stdout[-2500:], r.stderr[-1500:])
    assert r.stdout.count("max rel err vs reference") >= seeds  # (it did compare)


def test_whole_grid_datum_machinery_on_a_small_grid(tmp_path):
    """The whole-grid datum of the headline size (pattern: row lengths + digest; values: block digests of flux and
    bound_flux, the pressure field) is made by ``oracle/gen_golden_headline_pattern.py`` from a run of the reference on
    the whole grid and consumed by ``bench.whole_grid_check``.  Here the same two functions on a 3 072-cell grid, the
    reference run on the spot, the kernels through the host-emulation build: what the GPU test asserts at 1 971 054
    cells must hold here to rounding."""
    env = oracle.ref_env(extra_last=[ROOT])
    if env is None:
        pytest.skip("reference PorePy not present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden_headline_pattern.py"), "8", "2", str(tmp_path)],
                       env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
    fx = os.path.join(str(tmp_path), "headline_flux_pattern_8.npz")
    assert os.path.exists(fx), r.stderr[-2000:]
    import bench
    import porepy_amd as pa
    from oracle.gen_golden_headline_pattern import headline_digest
    from tests import _parity as P

    # (round 6: the fine datum -- 256-row blocks, sum |a| and max |a| of all six matrices -- by its own generator)
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden_headline_fine.py"), "8", "2", str(tmp_path)],
                        env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
    fx2 = os.path.join(str(tmp_path), "headline_fine_digest_8.npz")
    assert os.path.exists(fx2), r2.stderr[-2000:]
    out = bench.whole_grid_check(pa, 0, 1e-13, "amg", want_pattern=True, n_side=8, library=P.emulation_library(), fixture=fx,
                                 fine_fixture=fx2)
    fine = out["fine_values_vs_reference"]
    assert fine["rows_per_block"] == 256 and len([k for k in fine if k != "rows_per_block"]) == 6
    for k, d in fine.items():
        if k != "rows_per_block":
            assert d["sum_abs_worst_rel_diff"] < 1e-12 and d["max_abs_worst_rel_diff"] < 1e-12, (k, d)
    indptr, indices, rows, ref_digest = out.pop("_pattern")
    assert out["pattern_row_lengths_equal"] and headline_digest(indptr, indices, rows) == ref_digest
    v = out["values_vs_reference"]
    assert max(v["flux_worst_rel_diff_abs_sq_weighted"]) < 1e-12 and max(v["bound_flux_worst_rel_diff_abs_sq_weighted"]) < 1e-12
    for k in ("bound_pressure_cell", "bound_pressure_face", "vector_source", "bound_pressure_vector_source"):
        assert max(v[k + "_worst_rel_diff_abs_sq_weighted"]) < 1e-12, (k, v[k + "_worst_rel_diff_abs_sq_weighted"])
    assert v["pressure_norm_rel_diff"] < 1e-10 and v["pressure_block_squares_worst_rel_diff"] < 1e-9
