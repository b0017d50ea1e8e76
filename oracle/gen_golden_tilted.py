"""Golden fixtures for 2-D grids embedded in 3-D (rotated out of the xy-plane), made by
running the REFERENCE (numerics/fv/mpfa.py:733-754 rotation into the plane, :422-463 mapping
of the vector source back to the ambient space).

TEST INFRASTRUCTURE; build container only:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_tilted.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sps

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import KEYS, OUT, bc_vals, mixed_bc, pack_csr, perturb_interior  # noqa: E402
from oracle.ref_bridge import bc_to_raw, grid_to_raw  # noqa: E402


def rotation(axis, angle):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    W = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * W + (1 - np.cos(angle)) * W @ W


def save(name, g, R, rng, kinds, vdim=3):
    g.nodes = R @ g.nodes + np.array([[0.3], [-0.2], [0.7]])
    g.compute_geometry()
    nc = g.num_cells
    # full 3-D tensor, SPD
    B = rng.random((3, 3, nc)) - 0.5
    Kv = np.einsum("ikn,jkn->ijn", B, B) + 0.5 * np.eye(3)[:, :, None]
    K = pp.SecondOrderTensor(kxx=Kv[0, 0], kyy=Kv[1, 1], kzz=Kv[2, 2], kxy=Kv[0, 1], kxz=Kv[0, 2], kyz=Kv[1, 2])
    bc = mixed_bc(g, kinds)
    bv = bc_vals(g, bc, rng)
    gvec = rng.random(vdim * nc) - 0.5
    params = {"second_order_tensor": K, "bc": bc, "bc_values": bv, "mpfa_inverter": "python",
              "ambient_dimension": vdim, "vector_source": gvec}
    data = pp.initialize_data({}, "flow", params)
    d = pp.Mpfa("flow")
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
    store["vdim"] = np.int64(vdim)
    for k in KEYS:
        pack_csr("ref_" + k, data[pp.DISCRETIZATION_MATRICES]["flow"][k], store)
    pack_csr("ref_A", sps.csr_matrix(A), store)
    store["ref_rhs"] = b
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:32s} cells={nc:5d} {os.path.getsize(path)/1024:.0f} KiB")


def main():
    rng = np.random.default_rng(4242)
    g = pp.CartGrid([4, 3], [2.0, 1.0]); g.compute_geometry()
    save("tilted_cart2d_4x3", g, rotation([1, 2, 0.5], 0.9), rng, ["dir", "neu", "rob"])
    g = perturb_interior(pp.StructuredTriangleGrid([4, 4], [1, 1]), rng, 0.07)
    save("tilted_tri2d_4x4", g, rotation([0.2, -1, 0.4], 2.1), rng, ["dir", "neu"])
    # in the xy-plane but with a 3-D ambient space (zero z-columns in the vector source)
    g = perturb_interior(pp.StructuredTriangleGrid([3, 3], [1, 1]), rng, 0.07)
    save("tilted_flat_tri2d_3x3", g, np.eye(3), rng, ["dir", "neu", "rob"])
    # rotated out of the xy-plane with the DEFAULT ambient dimension (= 2): the reference un-rotates the vector-source
    # matrices with the leading corner of its block rotation (mpfa.py:425-462; the set-up of its own tests
    # test_mpfa.py:436-475) -- separate generator stream so that the fixtures above keep their bits
    rng2 = np.random.default_rng(4243)
    g = pp.CartGrid([4, 3], [2.0, 1.0]); g.compute_geometry()
    save("tilted_vdim2_cart2d_4x3", g, rotation([1, 2, 0.5], 0.9), rng2, ["dir", "neu", "rob"], vdim=2)
    g = perturb_interior(pp.StructuredTriangleGrid([4, 4], [1, 1]), rng2, 0.07)
    save("tilted_vdim2_tri2d_4x4", g, rotation([1, 0, 0], np.pi / 2), rng2, ["dir", "neu"], vdim=2)


if __name__ == "__main__":
    main()
