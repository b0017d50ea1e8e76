"""The numpy oracle's Biot coupling terms against the reference's pp.Biot output (fixtures made by
oracle/gen_golden_biot.py): pins the oracle that the device kernels are then checked against."""
import glob
import os

import pytest

from oracle import mpsa_oracle as so
from tests._golden import BIOT_KEYS, GOLDEN_DIR, BiotCase, rel_max_err

NAMES = [os.path.basename(p)[:-4] for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "biot_*.npz")))]


@pytest.mark.parametrize("name", NAMES)
def test_biot_oracle_matches_reference(name):
    c = BiotCase(name)
    out = so.discretize(c.grid, c.stiffness, c.bc, alphas=c.alphas, eta=c.eta_sub)  # (eta None: the default point)
    for k in ("stress", "bound_stress"):
        assert rel_max_err(out[k], c.ref_mech[k]) < 1e-10, (name, k)
    for k in BIOT_KEYS:
        for key in c.alphas:
            assert out[k][key].shape == c.ref[k][key].shape, (name, k, key)
            assert rel_max_err(out[k][key], c.ref[k][key]) < 1e-10, (name, k, key)
