"""MPSA golden fixtures with boundary conditions given in rotated / skewed face-wise bases
(tests/golden/mpsa_basis_*.npz), made by running the REFERENCE (params/bc.py:222-322,
numerics/fv/_fvutils.py:836-1083; setups after tests/numerics/fv/test_mpsa.py:735-860).

TEST INFRASTRUCTURE; build container only:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_mpsa_basis.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import perturb_interior  # noqa: E402
from oracle.gen_golden_mpsa import save_case  # noqa: E402
from oracle.gen_golden_mpsa_robin import robin_bc  # noqa: E402


def random_basis(g, rng):
    nd = g.dim
    B = np.zeros((nd, nd, g.num_faces))
    for f in range(g.num_faces):
        M = rng.random((nd, nd)) - 0.5
        q, _ = np.linalg.qr(M)
        B[:, :, f] = q + 0.15 * (rng.random((nd, nd)) - 0.5)  # rotated and slightly skewed
    return B


def main():
    rng = np.random.default_rng(616)
    g = perturb_interior(pp.StructuredTriangleGrid([3, 3], [1, 1]), rng, 0.08); nc = g.num_cells
    C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
    bc = robin_bc(g, rng, "mixed")
    bf = g.get_all_boundary_faces()
    east = bf[g.face_centers[0, bf] > 1 - 1e-9]   # rollers in the rotated basis: component 0 fixed, 1 free
    bc.is_dir[0, east] = True; bc.is_neu[0, east] = False; bc.is_rob[0, east] = False
    bc.is_dir[1, east] = False; bc.is_neu[1, east] = True; bc.is_rob[1, east] = False
    bc.basis = random_basis(g, rng)
    bv = (rng.random((2, g.num_faces)) - 0.4) * (bc.is_dir | bc.is_neu | bc.is_rob)
    extra = {"bc_is_rob": np.asarray(bc.is_rob, bool), "bc_robin_weight": bc.robin_weight, "bc_basis": bc.basis}
    save_case("mpsa_basis_tri2d_3x3_mixed", g, C, bc, bv.ravel("F"), rng.random(2 * nc) * 0.1, extra=extra)
    g = perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.08); nc = g.num_cells
    C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
    bc = robin_bc(g, rng, "mixed")
    bf = g.get_all_boundary_faces()
    east = bf[g.face_centers[0, bf] > 1 - 1e-9]
    bc.is_dir[0, east] = True; bc.is_neu[0, east] = False; bc.is_rob[0, east] = False
    bc.basis = random_basis(g, rng)
    bv = (rng.random((3, g.num_faces)) - 0.4) * (bc.is_dir | bc.is_neu | bc.is_rob)
    extra = {"bc_is_rob": np.asarray(bc.is_rob, bool), "bc_robin_weight": bc.robin_weight, "bc_basis": bc.basis}
    save_case("mpsa_basis_tet_2x2x2_mixed", g, C, bc, bv.ravel("F"), rng.random(3 * nc) * 0.05,
              keys=("stress", "bound_stress"), extra=extra)


if __name__ == "__main__":
    main()
