"""MPSA golden fixtures with boundary conditions given per SUB-FACE (tests/golden/mpsasub_*.npz), made by
running the REFERENCE: numerics/fv/mpsa.py:712-720 (sub-face conditions), :752-754 (no collapse of the stress
rows), :780-781 (boundary columns of the displacement reconstruction stay per sub-face), :1127-1138 (Neumann
data not divided by #nodes).  The reference has this branch but no test of it; the setup follows the MPFA
one (tests/numerics/fv/test_mpfa.py:1015-1040, oracle/gen_golden_subface.py).

TEST INFRASTRUCTURE; build container only:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_mpsa_subface.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

import porepy as pp
from porepy.numerics.fv import _fvutils

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import OUT, pack_csr, perturb_interior  # noqa: E402
from oracle.gen_golden_mpsa import KEYS, vec_bc  # noqa: E402
from oracle.ref_bridge import grid_to_raw  # noqa: E402


def save(name, g, C, bc_face, rng, robin=False, basis=False, hf_eta=None):
    # sub-face ids are positions in the stored face_nodes arrays: fix the storage order first so that the
    # reference numbers sub-faces like the raw (sorted) arrays of the fixture
    g.face_nodes.sort_indices()
    g.cell_faces.sort_indices()
    st = _fvutils.SubcellTopology(g)
    assert np.array_equal(st.subfno_unique, np.arange(st.num_subfno_unique))
    fn = g.face_nodes.tocsc()
    assert np.array_equal(st.fno_unique, np.repeat(np.arange(g.num_faces), np.diff(fn.indptr)))
    assert np.array_equal(st.nno_unique, fn.indices)
    bc = _fvutils.boundary_to_sub_boundary(bc_face, st)
    # make the sub-faces of one face differ: flip some Dirichlet components to Neumann (or Robin)
    for comp in range(g.dim):
        bsub = np.flatnonzero(bc.is_dir[comp])
        flip = bsub[rng.random(bsub.size) < 0.3]
        bc.is_dir[comp, flip] = False
        if robin:
            half = flip[: flip.size // 2]
            bc.is_neu[comp, half] = True
            bc.is_rob[comp, flip[flip.size // 2:]] = True
        else:
            bc.is_neu[comp, flip] = True
    if robin:
        w = 0.5 + rng.random(bc.robin_weight.shape[2])
        bc.robin_weight = np.einsum("ij,k->ijk", np.eye(g.dim), w)
    if basis:
        # a rotated (orthonormal) basis per sub-face (ExcludeBoundaries.basis_matrix, _fvutils.py:836-852, takes the
        # (nd, nd, Nsf) array as it is); sub-faces of one face get different ones
        nsub = bc.is_dir.shape[1]
        Q = np.empty((g.dim, g.dim, nsub))
        for k in range(nsub):
            q, r = np.linalg.qr(rng.standard_normal((g.dim, g.dim)))
            Q[:, :, k] = q * np.sign(np.diag(r))[None, :]
        bc.basis = Q
    stress, bound_stress, hf_cell, hf_bound = pp.Mpsa("mechanics")._stress_discretization(
        g, C, bc, eta=None, inverter="python", hf_eta=hf_eta)
    store = {}
    for k, v in grid_to_raw(g).items():
        store["grid_" + k] = np.asarray(v)
    for k in ("is_dir", "is_neu", "is_rob"):
        store["bc_" + k] = np.asarray(getattr(bc, k), bool)
    store["bc_robin_weight"] = np.asarray(bc.robin_weight, float)
    if basis:
        store["bc_basis"] = np.asarray(bc.basis, float)
    if hf_eta is not None:
        store["hf_eta"] = np.float64(hf_eta)  # reconstruction_eta (mpsa.py:757-761)
    store["stiffness"] = np.ascontiguousarray(C.values)
    for k, m in zip(KEYS, (stress, bound_stress, hf_cell, hf_bound)):
        pack_csr("ref_" + k, m, store)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:32s} cells={g.num_cells:5d} subfaces={st.num_subfno_unique:5d} "
          f"shapes={[m.shape for m in (stress, bound_stress, hf_cell, hf_bound)]}  {os.path.getsize(path)/1024:.0f} KiB")


def main():
    rng = np.random.default_rng(919)
    g = pp.CartGrid([4, 3]); g.compute_geometry(); nc = g.num_cells
    C = pp.FourthOrderTensor(1 + rng.random(nc), 0.5 + rng.random(nc))
    save("mpsasub_cart2d_4x3", g, C, vec_bc(g, "dir"), rng)
    g = perturb_interior(pp.StructuredTriangleGrid([3, 3], [1, 1]), rng, 0.08); nc = g.num_cells
    C = pp.FourthOrderTensor(1 + rng.random(nc), 0.5 + rng.random(nc))
    save("mpsasub_tri2d_3x3_rob", g, C, vec_bc(g, "dir"), rng, robin=True)
    g = perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.08); nc = g.num_cells
    C = pp.FourthOrderTensor(1 + rng.random(nc), 0.5 + rng.random(nc))
    save("mpsasub_tet3d_2x2x2", g, C, vec_bc(g, "roller"), rng)


def main_basis():
    """round 5: conditions per sub-face in a basis per sub-face"""
    rng = np.random.default_rng(920)
    g = perturb_interior(pp.StructuredTriangleGrid([3, 3], [1, 1]), rng, 0.08); nc = g.num_cells
    C = pp.FourthOrderTensor(1 + rng.random(nc), 0.5 + rng.random(nc))
    save("mpsasub_tri2d_3x3_basis_rob", g, C, vec_bc(g, "dir"), rng, robin=True, basis=True)
    g = perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.08); nc = g.num_cells
    C = pp.FourthOrderTensor(1 + rng.random(nc), 0.5 + rng.random(nc))
    save("mpsasub_tet3d_2x2x2_basis", g, C, vec_bc(g, "roller"), rng, basis=True)
    # ... and displacement traces reconstructed away from the continuity points (reconstruction_eta) with
    # conditions per sub-face
    g = perturb_interior(pp.StructuredTriangleGrid([3, 3], [1, 1]), rng, 0.08); nc = g.num_cells
    C = pp.FourthOrderTensor(1 + rng.random(nc), 0.5 + rng.random(nc))
    save("mpsasub_tri2d_3x3_hfeta_basis", g, C, vec_bc(g, "dir"), rng, basis=True, hf_eta=0.0)
    g = pp.CartGrid([3, 2, 2]); g.compute_geometry(); nc = g.num_cells
    C = pp.FourthOrderTensor(1 + rng.random(nc), 0.5 + rng.random(nc))
    save("mpsasub_cart3d_3x2x2_hfeta", g, C, vec_bc(g, "dir"), rng, hf_eta=0.25)


if __name__ == "__main__":
    if "basis" in sys.argv[1:]:
        main_basis()
    else:
        main()
