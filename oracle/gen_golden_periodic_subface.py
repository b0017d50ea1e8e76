"""MPFA golden fixtures with conditions per SUB-FACE on a grid WITH periodic faces (tests/golden/persub_*.npz), made by
running the REFERENCE: SubcellTopology gives the right sub-faces the numbers of the left ones and closes the gaps
(numerics/fv/_fvutils.py:91-160), a condition object with SubcellTopology.num_subfno_unique entries is taken per
sub-face (numerics/fv/mpfa.py:761-768), flux / bound_flux / the pressure traces keep one row per merged sub-face
(:1117-1125), the vector-source matrices are collapsed to faces with the rows of the left faces copied to the right
ones (:900-917).  The reference has both branches but no test of their combination.

TEST INFRASTRUCTURE; build container only:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_periodic_subface.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

import porepy as pp
from porepy.numerics.fv import _fvutils

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import KEYS, OUT, pack_csr  # noqa: E402
from oracle.gen_golden_periodic import faces_at, perturb_free_nodes  # noqa: E402
from oracle.ref_bridge import grid_to_raw  # noqa: E402


def save(name, g, pmap, rng, dir_axis):
    g.compute_geometry()
    g.face_nodes.sort_indices()
    g.cell_faces.sort_indices()
    pmap = np.asarray(pmap)
    g.set_periodic_map(pmap)
    nc = g.num_cells
    B = rng.random((3, 3, nc)) - 0.5
    Kv = np.einsum("ikn,jkn->ijn", B, B) + 0.5 * np.eye(3)[:, :, None]
    if g.dim == 2:
        Kv[2, :2] = Kv[:2, 2] = 0
    K = pp.SecondOrderTensor(kxx=Kv[0, 0], kyy=Kv[1, 1], kzz=Kv[2, 2], kxy=Kv[0, 1],
                             kxz=Kv[0, 2] if g.dim == 3 else None, kyz=Kv[1, 2] if g.dim == 3 else None)
    bf = g.get_all_boundary_faces()  # the periodic faces are not among them
    lo, hi = g.nodes[dir_axis].min(), g.nodes[dir_axis].max()
    xf = g.face_centers[dir_axis, bf]
    dirf = bf[(xf < lo + 1e-9) | (xf > hi - 1e-9)]
    bc_face = pp.BoundaryCondition(g, dirf, ["dir"] * dirf.size)
    st = _fvutils.SubcellTopology(g)
    bc = _fvutils.boundary_to_sub_boundary(bc_face, st)
    assert bc.is_dir.size == st.num_subfno_unique < g.face_nodes.nnz
    # make the sub-faces of one face differ
    bsub = np.flatnonzero(bc.is_dir)
    flip = bsub[rng.random(bsub.size) < 0.3]
    bc.is_dir[flip] = False
    half = flip[: flip.size // 2]
    bc.is_neu[half] = True
    bc.is_rob[flip[flip.size // 2:]] = True
    bc.robin_weight = 0.5 + rng.random(bc.robin_weight.shape)
    mats = pp.Mpfa("flow")._flux_discretization(g, K, bc, inverter="python", eta=None)
    store = {}
    for k, v in grid_to_raw(g).items():
        store["grid_" + k] = np.asarray(v)
    for k in ("is_dir", "is_neu", "is_rob", "is_internal"):
        store["bc_" + k] = np.asarray(getattr(bc, k), bool)
    store["bc_robin_weight"] = np.asarray(bc.robin_weight, float)
    store["periodic_face_map"] = pmap.astype(np.int64)
    store["perm"] = np.ascontiguousarray(K.values)
    store["fno_unique"] = np.asarray(st.fno_unique)
    store["nno_unique"] = np.asarray(st.nno_unique)
    for k, m in zip(KEYS, mats):
        pack_csr("ref_" + k, m, store)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:32s} cells={nc:5d} sub-faces {g.face_nodes.nnz} -> {st.num_subfno_unique} merged, shapes "
          f"{[m.shape for m in mats]}  {os.path.getsize(path)/1024:.0f} KiB")


def main():
    rng = np.random.default_rng(778)
    g = pp.CartGrid([4, 5], [1.0, 1.0]); g.compute_geometry()
    g = perturb_free_nodes(g, rng, 0.06, [1])
    save("persub_cart2d_4x5", g, np.vstack((faces_at(g, 1, 0.0), faces_at(g, 1, 1.0))), rng, 0)
    g = pp.StructuredTriangleGrid([4, 4], [1.0, 1.0]); g.compute_geometry()
    g = perturb_free_nodes(g, rng, 0.05, [1])
    save("persub_tri2d_4x4", g, np.vstack((faces_at(g, 1, 0.0), faces_at(g, 1, 1.0))), rng, 0)
    g = pp.StructuredTetrahedralGrid([2, 2, 3], [1.0, 1.0, 1.0]); g.compute_geometry()
    g = perturb_free_nodes(g, rng, 0.05, [2])
    save("persub_tet3d_2x2x3", g, np.vstack((faces_at(g, 2, 0.0), faces_at(g, 2, 1.0))), rng, 0)


if __name__ == "__main__":
    main()
