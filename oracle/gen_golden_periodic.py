"""Golden fixtures for grids with periodic faces (Grid.set_periodic_map, grids/grid.py:879-911),
made by running the REFERENCE: SubcellTopology merges the right sub-faces / nodes into the left ones
(numerics/fv/_fvutils.py:91-137), Mpfa copies the rows of the left faces to the right faces
(numerics/fv/mpfa.py:900-917); Tpfa pairs the cells across the periodic faces (numerics/fv/tpfa.py:114-262).

TEST INFRASTRUCTURE; build container only:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_periodic.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sps

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import KEYS, OUT, pack_csr  # noqa: E402
from oracle.ref_bridge import bc_to_raw, grid_to_raw  # noqa: E402


def faces_at(g, axis, value):
    return np.flatnonzero(np.abs(g.face_centers[axis] - value) < 1e-9)


def perturb_free_nodes(g, rng, rate, periodic_axes):
    """Move the nodes that are strictly inside the box; nodes on a periodic boundary keep their
    place, so the left and right faces stay translates of each other."""
    x = g.nodes.copy()
    d = g.dim
    lo, hi = x[:d].min(axis=1, keepdims=True), x[:d].max(axis=1, keepdims=True)
    inside = np.all((x[:d] > lo + 1e-9) & (x[:d] < hi - 1e-9), axis=0)
    x[:d, inside] += (rng.random((d, int(inside.sum()))) - 0.5) * rate
    g.nodes = x
    g.compute_geometry()
    return g


def save(name, g, pmap, rng, dir_axis, hetero=True, with_tpfa=True):
    g.compute_geometry()
    pmap = np.asarray(pmap)
    g.set_periodic_map(pmap)
    nc, nf = g.num_cells, g.num_faces
    if hetero:
        B = rng.random((3, 3, nc)) - 0.5
        Kv = np.einsum("ikn,jkn->ijn", B, B) + 0.5 * np.eye(3)[:, :, None]
        if g.dim == 2:
            Kv[2, :2] = Kv[:2, 2] = 0
        K = pp.SecondOrderTensor(kxx=Kv[0, 0], kyy=Kv[1, 1], kzz=Kv[2, 2], kxy=Kv[0, 1],
                                 kxz=Kv[0, 2] if g.dim == 3 else None, kyz=Kv[1, 2] if g.dim == 3 else None)
    else:
        K = pp.SecondOrderTensor(np.ones(nc))
    bf = g.get_all_boundary_faces()  # the periodic faces are not among them
    if dir_axis is None:
        bc = pp.BoundaryCondition(g)
    else:
        lo, hi = g.nodes[dir_axis].min(), g.nodes[dir_axis].max()
        xf = g.face_centers[dir_axis, bf]
        dirf = bf[(xf < lo + 1e-9) | (xf > hi - 1e-9)]
        bc = pp.BoundaryCondition(g, dirf, ["dir"] * dirf.size)
    bv = np.zeros(nf)
    bv[bf] = rng.random(bf.size) - 0.5
    gvec = rng.random(g.dim * nc) - 0.5
    params = {"second_order_tensor": K, "bc": bc, "bc_values": bv, "mpfa_inverter": "python", "vector_source": gvec}
    store = {}
    for k, v in grid_to_raw(g).items():
        store["grid_" + k] = np.asarray(v)
    for k, v in bc_to_raw(bc).items():
        store["bc_" + k] = v
    store["periodic_face_map"] = pmap.astype(np.int64)
    store["perm"] = np.ascontiguousarray(K.values)
    store["bc_values"] = bv
    store["vector_source_values"] = gvec
    for cls, tag in ((pp.Mpfa, "ref_"), (pp.Tpfa, "tpfa_")):
        if cls is pp.Tpfa and not with_tpfa:
            continue
        data = pp.initialize_data({}, "flow", dict(params))
        d = cls("flow")
        d.discretize(g, data)
        A, b = d.assemble_matrix_rhs(g, data)
        for k in KEYS:
            pack_csr(tag + k, data[pp.DISCRETIZATION_MATRICES]["flow"][k], store)
        pack_csr(tag + "A", sps.csr_matrix(A), store)
        store[tag + "rhs"] = b
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:32s} cells={nc:5d} {os.path.getsize(path)/1024:.0f} KiB")


def main():
    rng = np.random.default_rng(777)
    # the two set-ups of the reference's own tests (applications/test_utils/common_xpfa_tests.py:174-249)
    g = pp.CartGrid([3, 3]); g.compute_geometry()
    save("periodic_cart2d_3x3_both", g, [[0, 4, 8, 12, 13, 14], [3, 7, 11, 21, 22, 23]], rng, None, hetero=False)
    g = pp.CartGrid([5, 5]); g.compute_geometry()
    save("periodic_cart2d_5x5_y", g, np.vstack((faces_at(g, 1, 0.0), faces_at(g, 1, 5.0))), rng, 0, hetero=False)
    # heterogeneous anisotropic, perturbed interior
    g = pp.CartGrid([4, 5], [1.0, 1.0]); g.compute_geometry()
    g = perturb_free_nodes(g, rng, 0.06, [1])
    save("periodic_cart2d_4x5_aniso", g, np.vstack((faces_at(g, 1, 0.0), faces_at(g, 1, 1.0))), rng, 0)
    g = pp.StructuredTriangleGrid([4, 4], [1.0, 1.0]); g.compute_geometry()
    g = perturb_free_nodes(g, rng, 0.05, [1])
    save("periodic_tri2d_4x4", g, np.vstack((faces_at(g, 1, 0.0), faces_at(g, 1, 1.0))), rng, 0)
    g = pp.CartGrid([3, 3, 3], [1.0, 1.0, 1.0]); g.compute_geometry()
    g = perturb_free_nodes(g, rng, 0.05, [2])
    save("periodic_cart3d_3x3x3_z", g, np.vstack((faces_at(g, 2, 0.0), faces_at(g, 2, 1.0))), rng, 0)
    g = pp.StructuredTetrahedralGrid([2, 2, 3], [1.0, 1.0, 1.0]); g.compute_geometry()
    g = perturb_free_nodes(g, rng, 0.05, [2])
    save("periodic_tet3d_2x2x3_z", g, np.vstack((faces_at(g, 2, 0.0), faces_at(g, 2, 1.0))), rng, 0)


if __name__ == "__main__":
    main()
