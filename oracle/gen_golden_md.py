"""Golden fixtures for the subdomains of a MIXED-DIMENSIONAL grid: the reference's own Cartesian md mesher
builds a 3-D box cut by two intersecting planar fractures; every subdomain of dimension >= 2 is discretized by
the REFERENCE's pp.Mpfa with the boundary conditions its models give it (fracture faces tagged
`fracture_faces` are internal Neumann faces, mpfa.py:1452-1454; the 2-D fracture grids live in the planes
x = 0.5 and y = 0.5 of the 3-D space, ambient_dimension = 3, mpfa.py:733-754).  Same fixture layout as
gen_golden_tilted.py.

TEST INFRASTRUCTURE; build container only:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_md.py
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


def save(name, g, rng):
    nc = g.num_cells
    B = rng.random((3, 3, nc)) - 0.5
    Kv = np.einsum("ikn,jkn->ijn", B, B) + 0.5 * np.eye(3)[:, :, None]
    K = pp.SecondOrderTensor(kxx=Kv[0, 0], kyy=Kv[1, 1], kzz=Kv[2, 2], kxy=Kv[0, 1], kxz=Kv[0, 2], kyz=Kv[1, 2])
    # what the reference's models do: conditions on the domain boundary; BoundaryCondition itself marks the
    # fracture faces as internal Neumann faces
    bf = np.flatnonzero(g.tags["domain_boundary_faces"])
    x = g.face_centers[0, bf]
    kinds = ["dir" if (xi < 1e-9 or xi > 1 - 1e-9) else "neu" for xi in x]
    bc = pp.BoundaryCondition(g, bf, kinds)
    bv = np.zeros(g.num_faces)
    bv[bf] = rng.random(bf.size) - 0.3
    gvec = rng.random(3 * nc) - 0.5
    params = {"second_order_tensor": K, "bc": bc, "bc_values": bv, "mpfa_inverter": "python",
              "ambient_dimension": 3, "vector_source": gvec}
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
    for k in KEYS:
        pack_csr("ref_" + k, data[pp.DISCRETIZATION_MATRICES]["flow"][k], store)
    pack_csr("ref_A", sps.csr_matrix(A), store)
    store["ref_rhs"] = b
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:28s} dim={g.dim} cells={nc:4d} faces={g.num_faces:4d} internal faces={int(bc.is_internal.sum()):3d} "
          f"{os.path.getsize(path)/1024:.0f} KiB")


def main():
    rng = np.random.default_rng(99)
    f1 = np.array([[0.5, 0.5, 0.5, 0.5], [0.0, 1.0, 1.0, 0.0], [0.0, 0.0, 1.0, 1.0]])
    f2 = np.array([[0.0, 1.0, 1.0, 0.0], [0.5, 0.5, 0.5, 0.5], [0.0, 0.0, 1.0, 1.0]])
    mdg = pp.meshing.cart_grid([f1, f2], np.array([4, 4, 2]), physdims=np.array([1.0, 1.0, 1.0]))
    mdg.compute_geometry()
    n2 = 0
    for sd in mdg.subdomains():
        if sd.dim == 3:
            save("tilted_md_box_matrix3d", sd, rng)
        elif sd.dim == 2:
            save(f"tilted_md_box_fracture{n2}", sd, rng)
            n2 += 1


if __name__ == "__main__":
    main()
