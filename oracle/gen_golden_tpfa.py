"""Golden fixtures for the two-point flux approximation (numerics/fv/tpfa.py:84-279), made by
running the REFERENCE, including the 1-D grids its Mpfa delegates to Tpfa (mpfa.py:690-712).

TEST INFRASTRUCTURE; build container only:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_tpfa.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sps

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import KEYS, OUT, bc_vals, mixed_bc, pack_csr, perturb_interior  # noqa: E402
from oracle.gen_golden_tilted import rotation  # noqa: E402
from oracle.ref_bridge import bc_to_raw, grid_to_raw  # noqa: E402


def save(name, g, rng, kinds, vdim, via_mpfa=False):
    nc = g.num_cells
    B = rng.random((3, 3, nc)) - 0.5
    Kv = np.einsum("ikn,jkn->ijn", B, B) + 0.5 * np.eye(3)[:, :, None]
    K = pp.SecondOrderTensor(kxx=Kv[0, 0], kyy=Kv[1, 1], kzz=Kv[2, 2], kxy=Kv[0, 1], kxz=Kv[0, 2], kyz=Kv[1, 2])
    bc = mixed_bc(g, kinds)
    bv = bc_vals(g, bc, rng)
    gvec = rng.random(vdim * nc) - 0.5
    params = {"second_order_tensor": K, "bc": bc, "bc_values": bv, "ambient_dimension": vdim,
              "vector_source": gvec, "mpfa_inverter": "python"}
    data = pp.initialize_data({}, "flow", params)
    d = pp.Mpfa("flow") if via_mpfa else pp.Tpfa("flow")
    d.discretize(g, data)
    A, b = d.assemble_matrix_rhs(g, data)
    store = {}
    for k, v in grid_to_raw(g).items():
        store["grid_" + k] = np.asarray(v)
    for k, v in bc_to_raw(bc).items():
        store["bc_" + k] = v
    store["perm"] = np.ascontiguousarray(K.values)
    store["bc_values"] = bv
    store["vector_source_values"] = gvec
    store["vdim"] = np.array(vdim)
    store["via_mpfa"] = np.array(int(via_mpfa))
    for k in KEYS:
        pack_csr("ref_" + k, data[pp.DISCRETIZATION_MATRICES]["flow"][k], store)
    pack_csr("ref_A", sps.csr_matrix(A), store)
    store["ref_rhs"] = b
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:32s} cells={nc:5d} {os.path.getsize(path)/1024:.0f} KiB")


def main():
    rng = np.random.default_rng(99)
    g = pp.CartGrid(np.array([8]), np.array([2.0])); g.compute_geometry()
    save("tpfa_line_8", g, rng, ["dir", "neu"], 1)
    g = pp.CartGrid(np.array([6]), np.array([1.5]))
    g.nodes = rotation([1, -2, 0.7], 1.1) @ g.nodes + np.array([[0.1], [0.2], [0.3]])
    g.compute_geometry()
    save("tpfa_line_6_in_3d_via_mpfa", g, rng, ["dir", "dir"], 3, via_mpfa=True)
    g = pp.CartGrid([4, 3], [2.0, 1.0]); g.compute_geometry()
    save("tpfa_cart2d_4x3", g, rng, ["dir", "neu", "neu"], 2)
    g = perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.08)
    save("tpfa_tet3d_2x2x2", g, rng, ["dir", "neu"], 3)


if __name__ == "__main__":
    main()
