"""Differentiable two-point transmissibilities: the numpy oracle against what the reference's own forward AD
produced (tests/golden/tpfaad_*.npz, oracle/gen_golden_tpfa_ad.py), and the host-emulated kernel against
both."""
import numpy as np
import pytest

import porepy_amd as pa
from oracle import tpfa_ad_oracle as to
from tests import _parity as P

CASES = ["tpfaad_cart2d_4x3", "tpfaad_tri2d_3x3", "tpfaad_tet3d_2x2x2", "tpfaad_cart2d_tilted_3x2"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_ad(name):
    c = P.TpfaAdCase(name)
    t, jac, thi = to.transmissibility(c.grid, c.perm)
    # the reference numbers its half-faces face by face (sps.find), the oracle cell by cell
    fi = c.grid["cf_indices"]
    ci = np.repeat(np.arange(c.perm.shape[2]), np.diff(c.grid["cf_indptr"]))
    to_ref = np.lexsort((ci, fi))
    assert np.array_equal(fi[to_ref], c.hf_face) and np.array_equal(ci[to_ref], c.hf_cell)
    assert np.array_equal(c.grid["cf_sign"][to_ref], c.hf_sign)
    assert np.max(np.abs(thi[to_ref] - c.t_half_face_inv)) <= 1e-12 * np.max(np.abs(c.t_half_face_inv))
    assert np.max(np.abs(t - c.t_face)) <= 1e-12 * np.max(np.abs(c.t_face))
    assert abs(jac - c.dt_dk).max() <= 1e-12 * abs(c.dt_dk).max()


@pytest.mark.parametrize("name", CASES)
def test_emulated_kernel_matches_reference_ad(name):
    P.check_tpfa_ad_case(P.emulation_library(), name)


def test_chain_rule_against_finite_differences():
    """dt/dk against central differences of the oracle's t (the Jacobian is what Newton needs)."""
    c = P.TpfaAdCase("tpfaad_tet3d_2x2x2")
    t, jac, _ = to.transmissibility(c.grid, c.perm)
    rng = np.random.default_rng(0)
    dK = 1e-6 * (rng.random(c.perm.shape) - 0.5)
    tp, _, _ = to.transmissibility(c.grid, c.perm + dK)
    tm, _, _ = to.transmissibility(c.grid, c.perm - dK)
    nc = c.perm.shape[2]
    dk = np.ascontiguousarray(dK.reshape(9, nc).T).ravel()
    assert np.max(np.abs((tp - tm) / 2 - jac @ dk)) <= 1e-6 * np.max(np.abs(jac @ dk))


@pytest.mark.parametrize("name", ["tpfa_line_8", "tpfa_line_6_in_3d_via_mpfa"])
def test_one_dimensional_grids(name):
    """Fracture intersections of a mixed-dimensional model are 1-D grids (possibly embedded in 3-D): the
    kernel against the oracle on the grids of the TPFA fixtures."""
    import os

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    raw = {k[5:]: (z[k] if z[k].shape else z[k].item()) for k in z.files if k.startswith("grid_")}
    raw["dim"] = int(raw["dim"])
    g = pa.grid_from_raw(raw)
    t, jac = pa.DifferentiableTpfa(library=P.emulation_library()).transmissibility(g, z["perm"])
    t_o, jac_o, _ = to.transmissibility(raw, z["perm"])
    assert g.dim == 1
    assert np.max(np.abs(t - t_o)) <= 1e-12 * np.max(np.abs(t_o))
    assert abs(jac - jac_o).max() <= 1e-12 * abs(jac_o).max()
