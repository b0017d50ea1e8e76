"""MPSA golden fixtures with ``reconstruction_eta`` given PER SUB-FACE (tests/golden/mpsa_hfetasub_*.npz), made by running
the REFERENCE: ``Mpsa.discretize`` hands the parameter to ``_reconstruct_displacement`` (numerics/fv/mpsa.py:185, 757-761,
1187-1266), whose ``compute_dist_face_cell`` (numerics/fv/_fvutils.py:222-277) takes an array of
SubcellTopology.num_subfno_unique values as it is -- also on the boundary, where the scalar form uses 0.  The reference
has this branch but no test of it.

TEST INFRASTRUCTURE; build container only:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_mpsa_hfeta_sub.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gen_golden as gg  # noqa: E402
from oracle import gen_golden_mpsa as gm  # noqa: E402


def main():
    rng = np.random.default_rng(20260927)
    for name, g in (("mpsa_hfetasub_tri2d_3x3", gg.perturb_interior(pp.StructuredTriangleGrid([3, 3], [1, 1]), rng, 0.06)),
                    ("mpsa_hfetasub_tet3d_2x2x2", gg.perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.05))):
        g.face_nodes.sort_indices()  # sub-face numbering = sorted CSC positions (what the fixture's array follows)
        nd, nc, nf = g.dim, g.num_cells, g.num_faces
        C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
        bf = g.get_all_boundary_faces()
        kinds = np.array(["dir", "neu", "dir"])[np.arange(bf.size) % 3]
        bc = pp.BoundaryConditionVectorial(g, bf, list(kinds))
        bv = np.zeros((nd, nf))
        bv[:, bf] = rng.random((nd, bf.size)) - 0.5
        hf_sub = 0.02 + 0.4 * rng.random(g.face_nodes.nnz)
        gm.save_case(name, g, C, bc, bv.ravel("F"), rng.random(nd * nc) * np.repeat(g.cell_volumes, nd), eta=1.0 / 3.0,
                     extra={"hf_eta_sub": hf_sub}, more_params={"reconstruction_eta": hf_sub})


if __name__ == "__main__":
    main()
