"""The numpy restatement of the reference's differentiable MPFA flux (oracle/ad_flux_oracle.py) and the
device implementation (host-emulation build here; tests/test_gpu_parity.py runs the gfx950 library) against
fixtures the reference's own operator tree and forward AD produced (oracle/gen_golden_ad_flux.py)."""
import os

import numpy as np
import pytest
import scipy.sparse as sps

from oracle import ad_flux_oracle as ao
from tests import _parity as P

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ["adflux_unit_2cells", "adflux_unit_2cells_novs", "adflux_tet3d_2x2x2"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_ad(name):
    c = P.AdFluxCase(name)
    mats = {"flux": c.ref_mpfa_flux, "vector_source": c.ref_mpfa_vs}
    q, dq, J, r = ao.flux_system(c.grid, mats, c.perm, c.dk_dp, c.p, c.bc_flags, c.bc_values, c.vector_source, None)
    assert np.max(np.abs(q - c.ref_flux)) <= 1e-12 * np.max(np.abs(c.ref_flux))
    assert abs(dq - c.ref_flux_jac).max() <= 1e-12 * abs(c.ref_flux_jac).max()
    assert abs(J - c.ref_div_flux_jac).max() <= 1e-12 * abs(c.ref_div_flux_jac).max()
    assert np.max(np.abs(r - c.ref_div_flux)) <= 1e-12 * np.max(np.abs(c.ref_div_flux))


@pytest.mark.parametrize("name", CASES)
def test_emulated_device_path_reproduces_reference_ad(name):
    P.check_ad_flux_case(P.emulation_library(), name)
