"""Biot golden fixtures with the continuity points given PER SUB-FACE (tests/golden/biot_etasub_*.npz), made by running
the REFERENCE pp.Biot with ``mpsa_eta`` as an array of one value per sub-face (numerics/fv/biot.py:247-1135 hands eta to
the local discretization; numerics/fv/_fvutils.py:222-277 takes an array as it is, also on the boundary).  The reference
has no test of the array form with the coupling terms.

TEST INFRASTRUCTURE; build container only:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_biot_etasub.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import perturb_interior  # noqa: E402
from oracle.gen_golden_biot import save  # noqa: E402
from oracle.gen_golden_mpsa_robin import robin_bc  # noqa: E402


def main():
    rng = np.random.default_rng(809)
    g = perturb_interior(pp.StructuredTriangleGrid([3, 3], [1, 1]), rng, 0.08); nc = g.num_cells
    g.face_nodes.sort_indices()  # sub-face numbering = sorted CSC positions (what the fixture's array follows)
    C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
    bc = robin_bc(g, rng, "mixed")
    a2 = pp.SecondOrderTensor(kxx=1 + rng.random(nc), kyy=0.5 + rng.random(nc), kxy=0.3 * rng.random(nc))
    eta_sub = 0.05 + 0.35 * rng.random(g.face_nodes.nnz)
    save("biot_etasub_tri2d_3x3", g, C, bc, {"pressure": 0.8, "temperature": a2},
         more_params={"mpsa_eta": eta_sub}, extra={"eta_sub": eta_sub})
    g = perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.08); nc = g.num_cells
    g.face_nodes.sort_indices()
    C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
    bc = robin_bc(g, rng, "mixed")
    a3 = pp.SecondOrderTensor(kxx=1 + rng.random(nc), kyy=0.5 + rng.random(nc), kzz=0.7 + rng.random(nc),
                              kxy=0.2 * rng.random(nc), kxz=0.1 * rng.random(nc), kyz=0.15 * rng.random(nc))
    eta_sub = 0.05 + 0.35 * rng.random(g.face_nodes.nnz)
    save("biot_etasub_tet_2x2x2", g, C, bc, {"pressure": a3}, more_params={"mpsa_eta": eta_sub}, extra={"eta_sub": eta_sub})


if __name__ == "__main__":
    main()
