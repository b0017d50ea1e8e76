"""CPU restatement (numpy) of the differentiable two-point transmissibilities of the reference -
TEST INFRASTRUCTURE, never imported by the product.

Follows AdTpfaFlux.__transmissibility_matrix (models/constitutive_laws.py:1504-1578) on the half-face
geometry of DifferentiableTpfa (numerics/fv/tpfa.py): _normal_vectors (:520-574, rows 3 hf + r hold n_f at
the columns 9 c + 3 r + s), _cell_face_vectors (:466-518, d = x_f - x_c), _cell_face_distances (:576-596, the
SQUARED distance), half_face_map with sign (:402-464).  Pinned by tests/golden/tpfaad_*.npz, which the
reference's own forward AD produced (oracle/gen_golden_tpfa_ad.py).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps


def transmissibility(raw: dict, perm: np.ndarray):
    """(t_f (Nf,), dt_f/dk_c csr (Nf, 9 Nc), 1 / t_half_face (n_hf,)) for perm of shape (3, 3, Nc)."""
    nc = raw["cell_centers"].shape[1]
    nf = raw["face_centers"].shape[1]
    fi = raw["cf_indices"].astype(np.int64)
    ci = np.repeat(np.arange(nc), np.diff(raw["cf_indptr"]))
    sgn = raw["cf_sign"].astype(float)
    d = raw["face_centers"][:, fi] - raw["cell_centers"][:, ci]         # (3, n_hf)
    n = raw["face_normals"][:, fi]
    dist = (d * d).sum(axis=0)
    t_hf = np.einsum("re,rse,se->e", d, perm[:, :, ci], n) / dist      # d^T K n / |d|^2
    t_hf_inv = 1.0 / t_hf
    s = np.bincount(fi, weights=sgn * t_hf_inv, minlength=nf)
    t_f = 1.0 / s
    # d t_f / d K_c[r][s] = t_f^2 sgn (1 / t_hf^2) d_r n_s / |d|^2
    w = t_f[fi] ** 2 * sgn * t_hf_inv ** 2 / dist
    vals = (w[None, None, :] * d[:, None, :] * n[None, :, :]).reshape(9, -1).T   # (n_hf, 9): 3 r + s
    rows = np.repeat(fi, 9)
    cols = (9 * ci[:, None] + np.arange(9)[None, :]).ravel()
    jac = sps.csr_matrix((vals.ravel(), (rows, cols)), shape=(nf, 9 * nc))
    return t_f, jac, t_hf_inv
