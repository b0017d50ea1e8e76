"""Golden fixtures for the differentiable two-point transmissibilities: the face transmissibility
``t_f_full`` of AdTpfaFlux.__transmissibility_matrix (models/constitutive_laws.py:1504-1578) with its
Jacobian with respect to the cell-wise permeability entries, evaluated by the REFERENCE's own forward AD
(pp.ad.AdArray) on the REFERENCE's DifferentiableTpfa geometry matrices (numerics/fv/tpfa.py:281-750).

TEST INFRASTRUCTURE; build container only:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_tpfa_ad.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sps

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import OUT, pack_csr, perturb_interior  # noqa: E402
from oracle.gen_golden_tilted import rotation  # noqa: E402
from oracle.ref_bridge import grid_to_raw  # noqa: E402


def save(name, g, rng):
    nc = g.num_cells
    B = rng.random((3, 3, nc)) - 0.5
    Kv = np.einsum("ikn,jkn->ijn", B, B) + 0.5 * np.eye(3)[:, :, None]
    # k_c of the reference: 9 entries per cell, cell-major, K[r][s] at 3 r + s (constitutive_laws.py:1537-1541)
    k_c = pp.ad.AdArray(np.ascontiguousarray(Kv.reshape(9, nc).T).ravel(), sps.identity(9 * nc, format="csr"))
    diff = pp.numerics.fv.tpfa.DifferentiableTpfa()
    n, d_vec, dist = diff.half_face_geometry_matrices([g])
    d_n_by_dist = sps.diags(1 / dist) * d_vec @ n
    t_hf_inv = 1 / (sps.csr_matrix(d_n_by_dist) @ k_c)
    hf_to_f = diff.half_face_map([g], to_entity="faces", with_sign=True)
    t_f = 1 / (sps.csr_matrix(hf_to_f) @ t_hf_inv)
    store = {}
    for k, v in grid_to_raw(g).items():
        store["grid_" + k] = np.asarray(v)
    store["perm"] = np.ascontiguousarray(Kv)
    store["ref_t_face"] = np.asarray(t_f.val)
    pack_csr("ref_dt_dk", sps.csr_matrix(t_f.jac), store)
    # the order of the half-faces the reference works in (sps.find of cell_faces)
    fi, ci, sgn = sps.find(g.cell_faces)
    store["ref_hf_face"], store["ref_hf_cell"], store["ref_hf_sign"] = fi, ci, sgn
    store["ref_t_half_face_inv"] = np.asarray(t_hf_inv.val)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:32s} cells={nc:5d} faces={g.num_faces:5d} {os.path.getsize(path)/1024:.0f} KiB")


def main():
    rng = np.random.default_rng(2024)
    g = pp.CartGrid([4, 3], [2.0, 1.0]); g.compute_geometry()
    save("tpfaad_cart2d_4x3", g, rng)
    g = perturb_interior(pp.StructuredTriangleGrid([3, 3], [1, 1]), rng, 0.06)
    save("tpfaad_tri2d_3x3", g, rng)
    g = perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.08)
    save("tpfaad_tet3d_2x2x2", g, rng)
    g = pp.CartGrid([3, 2], [1.5, 1.0])
    g.nodes = rotation([1, -2, 0.7], 1.1) @ g.nodes + np.array([[0.1], [0.2], [0.3]])
    g.compute_geometry()
    save("tpfaad_cart2d_tilted_3x2", g, rng)


if __name__ == "__main__":
    main()
