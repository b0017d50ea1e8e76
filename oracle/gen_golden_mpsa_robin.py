"""MPSA golden fixtures with Robin boundary conditions (tests/golden/mpsa_robin_*.npz) made by
running the REFERENCE (numerics/fv/mpsa.py:784-930 local systems, :1381-1459 Robin displacement
rows, :1932-2000 corner rule; Robin setups as in tests/numerics/fv/test_mpsa.py:480-800).

TEST INFRASTRUCTURE; build container only:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_mpsa_robin.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import perturb_interior  # noqa: E402
from oracle.gen_golden_mpsa import save_case  # noqa: E402


def robin_bc(g, rng, pattern):
    nd = g.dim
    bf = g.get_all_boundary_faces()
    fc = g.face_centers
    bc = pp.BoundaryConditionVectorial(g)
    bot = bf[fc[nd - 1, bf] < 1e-9]
    top = bf[fc[nd - 1, bf] > fc[nd - 1].max() - 1e-9]
    west = bf[fc[0, bf] < 1e-9]
    east = bf[fc[0, bf] > fc[0].max() - 1e-9]
    if pattern == "all_rob":
        bc.is_rob[:, bf] = True
        bc.is_neu[:, bf] = False
    else:  # mixed: Dirichlet bottom, Robin top (all components), Robin in one component west, Neumann rest
        bc.is_dir[:, bot] = True
        bc.is_neu[:, bot] = False
        bc.is_rob[:, top] = True
        bc.is_neu[:, top] = False
        w_only = np.setdiff1d(west, np.r_[bot, top])
        bc.is_rob[0, w_only] = True
        bc.is_neu[0, w_only] = False
    W = np.zeros((nd, nd, g.num_faces))
    for f in range(g.num_faces):
        B = rng.random((nd, nd)) - 0.5
        W[:, :, f] = B @ B.T + (0.5 + rng.random()) * np.eye(nd)
    bc.robin_weight = W
    return bc


def main():
    rng = np.random.default_rng(515)
    g = perturb_interior(pp.StructuredTriangleGrid([3, 3], [1, 1]), rng, 0.08)
    nc = g.num_cells
    C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
    for pattern in ("mixed", "all_rob"):
        bc = robin_bc(g, rng, pattern)
        bv = (rng.random((2, g.num_faces)) - 0.4) * (bc.is_dir | bc.is_neu | bc.is_rob)
        save_case(f"mpsa_robin_tri2d_3x3_{pattern}", g, C, bc, bv.ravel("F"), rng.random(2 * nc) * 0.1,
                  extra={"bc_is_rob": np.asarray(bc.is_rob, bool), "bc_robin_weight": bc.robin_weight})
    g = pp.CartGrid([3, 3]); g.compute_geometry(); nc = g.num_cells
    C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
    bc = robin_bc(g, rng, "all_rob")
    bv = (rng.random((2, g.num_faces)) - 0.4) * bc.is_rob
    save_case("mpsa_robin_cart2d_3x3_all_rob", g, C, bc, bv.ravel("F"), rng.random(2 * nc) * 0.1,
              extra={"bc_is_rob": np.asarray(bc.is_rob, bool), "bc_robin_weight": bc.robin_weight})
    g = perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.08); nc = g.num_cells
    C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
    bc = robin_bc(g, rng, "mixed")
    bv = (rng.random((3, g.num_faces)) - 0.4) * (bc.is_dir | bc.is_neu | bc.is_rob)
    save_case("mpsa_robin_tet_2x2x2_mixed", g, C, bc, bv.ravel("F"), rng.random(3 * nc) * 0.05,
              keys=("stress", "bound_stress"),
              extra={"bc_is_rob": np.asarray(bc.is_rob, bool), "bc_robin_weight": bc.robin_weight})


if __name__ == "__main__":
    main()
