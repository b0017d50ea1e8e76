"""Golden fixtures with CONTINUITY POINTS PER SUB-FACE (tests/golden/etasub_*.npz, mpsa_etasub_*.npz) by running the
REFERENCE PorePy.   TEST INFRASTRUCTURE; build container only:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo python /root/repo/oracle/gen_golden_etasub.py
``mpfa_eta`` / ``mpsa_eta`` given as arrays of SubcellTopology.num_subfno_unique values (numerics/fv/_fvutils.py:222-277,
mpfa.py:599-609, mpsa.py:293-303, 647-652): every sub-face its own continuity point, used as given also on the
boundary.  The reference's own tests never exercise the array form.  Also: ``reconstruction_eta`` (mpsa_hfeta_*)."""
from __future__ import annotations

import os
import sys

import numpy as np

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gen_golden as gg  # noqa: E402
from oracle import gen_golden_mpsa as gm  # noqa: E402


def main():
    rng = np.random.default_rng(20260926)
    # --- MPFA, 2-D triangles and 3-D tetrahedra
    for name, g in (("etasub_tri2d_4x4", gg.perturb_interior(pp.StructuredTriangleGrid([4, 4], [1, 1]), rng, 0.06)),
                    ("etasub_tet3d_2x2x2", gg.perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.05))):
        g.face_nodes.sort_indices()  # sub-face numbering = sorted CSC positions (what the fixture's eta_sub follows)
        nc = g.num_cells
        kw = dict(kxx=1 + rng.random(nc), kyy=2 + rng.random(nc), kxy=0.3 * rng.random(nc))
        if g.dim == 3:
            kw.update(kzz=0.5 + rng.random(nc), kxz=0.1 * rng.random(nc), kyz=0.1 * rng.random(nc))
        K = pp.SecondOrderTensor(**kw)
        bc = gg.mixed_bc(g, ["dir", "neu", "rob"])
        eta_sub = 0.05 + 0.35 * rng.random(g.face_nodes.nnz)
        gg.save_case(name, g, K, bc, gg.bc_vals(g, bc, rng), rng.random(nc) * g.cell_volumes, eta=eta_sub,
                     extra={"eta": np.array(np.nan), "eta_sub": eta_sub})
    # --- MPSA
    for name, g in (("mpsa_etasub_tri2d_3x3", gg.perturb_interior(pp.StructuredTriangleGrid([3, 3], [1, 1]), rng, 0.06)),
                    ("mpsa_etasub_tet3d_2x2x2", gg.perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.05))):
        g.face_nodes.sort_indices()
        nd, nc, nf = g.dim, g.num_cells, g.num_faces
        C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
        bf = g.get_all_boundary_faces()
        kinds = np.array(["dir", "neu", "dir"])[np.arange(bf.size) % 3]
        bc = pp.BoundaryConditionVectorial(g, bf, list(kinds))
        bv = np.zeros((nd, nf))
        bv[:, bf] = rng.random((nd, bf.size)) - 0.5
        eta_sub = 0.05 + 0.35 * rng.random(g.face_nodes.nnz)
        gm.save_case(name, g, C, bc, bv.ravel("F"), rng.random(nd * nc) * np.repeat(g.cell_volumes, nd), eta=eta_sub,
                     extra={"eta": np.array(np.nan), "eta_sub": eta_sub})
        # --- the same problem with ``reconstruction_eta`` different from a scalar ``mpsa_eta`` (mpsa.py:185, 757-761):
        # displacement traces reconstructed at x_f + 0.1 (x_v - x_f)
        gm.save_case(name.replace("etasub", "hfeta"), g, C, bc, bv.ravel("F"),
                     rng.random(nd * nc) * np.repeat(g.cell_volumes, nd), eta=1.0 / 3.0,
                     extra={"hf_eta": np.array(0.1)}, more_params={"reconstruction_eta": 0.1})


if __name__ == "__main__":
    main()
