"""CPU restatement (numpy / scipy) of the reference's differentiable MPFA flux - TEST INFRASTRUCTURE, never
imported by the product.

Follows AdTpfaFlux.diffusive_flux with an Mpfa base discretization (models/constitutive_laws.py:1195-1336)
and its product-rule functions __mpfa_flux_discretization (:1580-1625) and
__mpfa_vector_source_discretization (:1627-1721) on one subdomain without interfaces:

    q  = T_MPFA p + t_bnd bc + VS_MPFA g
    dq = T_MPFA dp + diag(p_diff + g_diff) d t_f + diag(bc) d t_bnd

with t_f the two-point transmissibilities filtered on Neumann / internal faces (:1256-1259),
t_bnd = neu sgn - dir sgn t_f (:1263-1272), p_diff = face_pairing_from_cell_array p
(numerics/fv/tpfa.py:661-681), g_diff = hf_to_f d_vec cells_to_hf g (:1294-1305).  Pinned by
tests/golden/adflux_*.npz, which the reference's own operator tree and forward AD produced
(oracle/gen_golden_ad_flux.py)."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps

from oracle import tpfa_ad_oracle as to


def flux_system(raw: dict, mats: dict, perm: np.ndarray, dk_dp, p, bc_flags, bc_values, vector_source=None, source=None):
    """(q (Nf,), dq/dp csr (Nf, Nc), J = d(div q)/dp csr (Nc, Nc), r = div q - source (Nc,)).
    mats: the MPFA matrices for `perm` ("flux", "vector_source"); dk_dp: (3, 3, Nc) or None; bc_flags: per face
    1 Dirichlet / 2 Neumann / 8 internal (the C ABI's flag byte)."""
    nc = raw["cell_centers"].shape[1]
    nf = raw["face_centers"].shape[1]
    nd = int(raw["dim"])
    fi = raw["cf_indices"].astype(np.int64)
    ci = np.repeat(np.arange(nc), np.diff(raw["cf_indptr"]))
    sgn = raw["cf_sign"].astype(float)
    t_f, dt_dk, _ = to.transmissibility(raw, perm)
    if dk_dp is None:
        dt_dp = sps.csr_matrix((nf, nc))
    else:
        dk = np.asarray(dk_dp, dtype=float).reshape(9, nc)          # row 3 r + s
        chain = sps.csr_matrix((dk.T.ravel(), (np.repeat(9 * np.arange(nc), 9) + np.tile(np.arange(9), nc),
                                               np.repeat(np.arange(nc), 9))), shape=(9 * nc, nc))
        dt_dp = (dt_dk @ chain).tocsr()
    sides = np.bincount(fi, minlength=nf)
    bnd = sides == 1
    flags = np.asarray(bc_flags)
    internal = (flags & 8) != 0
    is_dir = bnd & ((flags & 1) != 0) & ~internal
    is_neu = bnd & ~is_dir & ~internal
    filt = np.where(bnd & ~is_dir, 0.0, 1.0)
    bsgn = np.zeros(nf)
    one_sided = bnd[fi]
    bsgn[fi[one_sided]] = sgn[one_sided]
    pair = sps.csr_matrix((sgn, (fi, ci)), shape=(nf, nc))           # face_pairing_from_cell_array
    p_diff = pair @ p
    g_diff = np.zeros(nf)
    if vector_source is not None:
        g = np.asarray(vector_source, dtype=float).reshape(nc, nd)
        d = raw["face_centers"][:nd, fi] - raw["cell_centers"][:nd, ci]
        g_diff = np.bincount(fi, weights=sgn * np.einsum("re,er->e", d, g[ci]), minlength=nf)
    bc = np.zeros(nf) if bc_values is None else np.asarray(bc_values, dtype=float)
    t_filt = filt * t_f
    t_bnd = is_neu * bsgn - is_dir * bsgn * t_filt
    q = mats["flux"] @ p + t_bnd * bc
    if vector_source is not None:
        q = q + mats["vector_source"] @ np.asarray(vector_source, dtype=float)
    w = filt * (p_diff + g_diff) - is_dir * bsgn * bc
    dq = (mats["flux"] + sps.diags(w) @ dt_dp).tocsr()
    div = sps.csr_matrix((sgn, (ci, fi)), shape=(nc, nf))
    J = (div @ dq).tocsr()
    r = div @ q - (0.0 if source is None else np.asarray(source, dtype=float))
    return q, dq, J, r
