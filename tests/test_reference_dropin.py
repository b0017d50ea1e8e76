"""Drop-in check against the reference package itself (build container only; skipped where
/root/reference is absent): an unmodified PorePy model runs with ``pp.Mpfa`` rebound."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference PorePy not present")
def test_single_phase_flow_model_with_rebound_mpfa():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "oracle", "shim"), REF, ROOT])
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_dropin_script.py")], env=env, cwd="/tmp",
                       capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stderr[-2000:]
    out = json.loads(line[-1][7:])
    assert out["cells"] == 2500
    assert out["calls_into_device_path"] >= 1
    assert out["p_rel_err"] < 1e-10
    assert out["A_rel_err"] < 1e-10  # (the final residual vector is round-off in both runs)
    assert out["p_rel_err_hip_solver"] < 1e-10 and out["hip_solver_iterations"] > 0
    assert abs(out["p_sum_ref"] - 8750.0) < 1e-6  # config C1 of the reference


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference PorePy not present")
def test_momentum_balance_and_poromechanics_models_with_rebound_mpsa_biot():
    """pp.Mpsa, pp.Biot and pp.Mpfa rebound under the reference's own MomentumBalance and
    Poromechanics models: displacement / pressure and the Jacobian reproduce the untouched runs."""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "oracle", "shim"), REF, ROOT])
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_dropin_mech_script.py")], env=env, cwd="/tmp",
                       capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stderr[-2000:]
    out = json.loads(line[-1][7:])
    assert out["calls"]["mpsa"] >= 1 and out["calls"]["biot"] >= 1
    assert out["mech_dofs"] == 72 and out["poro_dofs"] == 108
    assert out["mech_x_rel_err"] < 1e-10 and out["mech_A_rel_err"] < 1e-10
    assert out["poro_x_rel_err"] < 1e-10 and out["poro_A_rel_err"] < 1e-10


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference PorePy not present")
def test_differentiable_tpfa_transmissibilities_in_the_reference_model():
    """The reference's unit-test model of its differentiable TPFA flux (two cells, pressure-dependent
    full-tensor permeability): ``AdTpfaFlux.__transmissibility_matrix`` through its operator tree and
    forward AD against pfv_tpfa_transmissibility_ad chained with the Jacobian of k_c."""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "oracle", "shim"), REF])
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_dropin_adtpfa_script.py")], env=env, cwd="/tmp",
                       capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stderr[-2000:]
    out = json.loads(line[-1][7:])
    for base in ("tpfa", "mpfa"):
        o = out[base]
        assert o["faces"] == 7 and o["dofs"] == 2 and o["jac_nnz_ref"] > 0
        assert o["t_rel_err"] < 1e-12 and o["jac_rel_err"] < 1e-12
        # the mixin of porepy_amd.as_porepy_ad_tpfa_flux() in front of the reference's classes: flux and
        # pressure trace of the whole model, values and Jacobians
        m = out[base + "_model"]
        assert m["device_path_calls"] >= 1 and m["flux_jac_nnz"] > 0
        assert max(m["flux_rel_err"], m["flux_jac_rel_err"], m["trace_rel_err"], m["trace_jac_rel_err"]) < 1e-12


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference PorePy not present")
def test_mixed_dimensional_model_with_rebound_mpfa_and_hip_solver():
    """The structured stand-in for BASELINE configs[4]: the reference's SinglePhaseFlow on a 3-D box with two
    intersecting fractures (2-D subdomains, 1-D intersection, mortar grids), pp.Mpfa rebound and the coupled
    Jacobian solved by the HIP Krylov solver: all unknowns (matrix, fractures, intersection, mortar fluxes)
    and the Jacobian reproduce the untouched run."""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "oracle", "shim"), REF, ROOT])
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_dropin_md_script.py")], env=env, cwd="/tmp",
                       capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stderr[-2000:]
    out = json.loads(line[-1][7:])
    assert out["dims"] == [3, 2, 1] and out["subdomains"] == 4 and out["interfaces"] == 4 and out["mortar_cells"] > 0
    assert out["device_calls_by_dim"]["3"] >= 1 and out["device_calls_by_dim"]["2"] >= 2
    assert out["x_rel_err"] < 1e-10 and out["A_rel_err"] < 1e-10
    assert out["x_rel_err_hip_solver"] < 1e-10 and out["hip_solver_iterations"] > 0
