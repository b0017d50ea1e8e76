"""MPFA golden fixtures with boundary conditions given per SUB-FACE (tests/golden/subface_*.npz),
made by running the REFERENCE (numerics/fv/mpfa.py:761-768 sub-face conditions, :1117-1125 no
collapse of the flux rows, :1516-1523 Neumann data not divided by #nodes; setup after
tests/numerics/fv/test_mpfa.py:1015-1040).

TEST INFRASTRUCTURE; build container only:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_subface.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

import porepy as pp
from porepy.numerics.fv import _fvutils

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import KEYS, OUT, mixed_bc, pack_csr, perturb_interior  # noqa: E402
from oracle.ref_bridge import grid_to_raw  # noqa: E402


def save(name, g, K, bc_face, rng):
    # sub-face ids are positions in the stored face_nodes arrays: fix the storage order first so that
    # the reference numbers sub-faces like the raw (sorted) arrays of the fixture
    g.face_nodes.sort_indices()
    g.cell_faces.sort_indices()
    st = _fvutils.SubcellTopology(g)
    assert np.array_equal(st.subfno_unique, np.arange(st.num_subfno_unique))
    bc = _fvutils.boundary_to_sub_boundary(bc_face, st)
    # make the sub-faces of one face differ: flip some Dirichlet sub-faces to Neumann / Robin
    bsub = np.flatnonzero(bc.is_dir)
    flip = bsub[rng.random(bsub.size) < 0.3]
    bc.is_dir[flip] = False
    half = flip[: flip.size // 2]
    bc.is_neu[half] = True
    bc.is_rob[flip[flip.size // 2:]] = True
    bc.robin_weight = 0.5 + rng.random(bc.robin_weight.shape)
    # the sub-face ordering of the condition object must be the face_nodes CSC order
    fn = g.face_nodes.tocsc()
    assert np.array_equal(st.fno_unique, np.repeat(np.arange(g.num_faces), np.diff(fn.indptr)))
    assert np.array_equal(st.nno_unique, fn.indices)
    mats = pp.Mpfa("flow")._flux_discretization(g, K, bc, inverter="python", eta=None)
    store = {}
    for k, v in grid_to_raw(g).items():
        store["grid_" + k] = np.asarray(v)
    for k in ("is_dir", "is_neu", "is_rob", "is_internal"):
        store["bc_" + k] = np.asarray(getattr(bc, k), bool)
    store["bc_robin_weight"] = np.asarray(bc.robin_weight, float)
    store["perm"] = np.ascontiguousarray(K.values)
    for k, m in zip(KEYS, mats):
        pack_csr("ref_" + k, m, store)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:32s} cells={g.num_cells:5d} subfaces={st.num_subfno_unique:5d}  {os.path.getsize(path)/1024:.0f} KiB")


def main():
    rng = np.random.default_rng(717)
    g = pp.CartGrid([4, 3]); g.compute_geometry(); nc = g.num_cells
    K = pp.SecondOrderTensor(kxx=1 + rng.random(nc), kyy=2 + rng.random(nc), kxy=0.3 * rng.random(nc))
    save("subface_cart2d_4x3", g, K, mixed_bc(g, ["dir", "dir", "neu"]), rng)
    g = perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.08); nc = g.num_cells
    k = 1 + rng.random(nc)
    K = pp.SecondOrderTensor(kxx=k, kyy=2 * k, kzz=0.5 * k, kxy=0.2 * k, kxz=0.05 * k, kyz=0.1 * k)
    save("subface_tet3d_2x2x2", g, K, mixed_bc(g, ["dir", "dir", "rob", "neu"]), rng)


if __name__ == "__main__":
    main()
