"""Golden fixtures for partial discretization / update_discretization, made by running
the REFERENCE (numerics/fv/mpfa.py:169-204,466-590; _fvutils.py:1090-1462).

TEST INFRASTRUCTURE; build container only:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_partial.py
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import KEYS, OUT, mixed_bc, pack_csr, perturb_interior  # noqa: E402
from oracle.ref_bridge import bc_to_raw, grid_to_raw  # noqa: E402


def discretize(g, K, bc, extra=None, data=None):
    params = {"second_order_tensor": K, "bc": bc, "mpfa_inverter": "python"}
    params.update(extra or {})
    if data is None:
        data = pp.initialize_data({}, "flow", params)
    else:
        data[pp.PARAMETERS]["flow"].update(params)
    pp.Mpfa("flow").discretize(g, data)
    return data


def save(name, g, K, bc, specs, K_new, modified_cells):
    store = {}
    for k, v in grid_to_raw(g).items():
        store["grid_" + k] = np.asarray(v)
    for k, v in bc_to_raw(bc).items():
        store["bc_" + k] = v
    store["perm"] = np.ascontiguousarray(K.values)
    store["num_partial"] = np.array(len(specs))
    for i, spec in enumerate(specs):
        data = discretize(g, K, bc, spec)
        pd = data[pp.PARAMETERS]["flow"]
        for kind in ("cells", "faces", "nodes"):
            store[f"p{i}_spec_{kind}"] = np.asarray(spec.get("specified_" + kind, np.array([-1])), dtype=np.int64)
        store[f"p{i}_active_faces"] = np.asarray(pd["active_faces"], dtype=np.int64)
        store[f"p{i}_active_cells"] = np.asarray(pd["active_cells"], dtype=np.int64)
        for k in KEYS:
            pack_csr(f"p{i}_{k}", data[pp.DISCRETIZATION_MATRICES]["flow"][k], store)
    # update: full discretization with K, then new permeability in a few cells
    data = discretize(g, K, bc)
    data[pp.PARAMETERS]["flow"]["second_order_tensor"] = K_new
    data["update_discretization"] = {"modified_cells": np.asarray(modified_cells)}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pp.Mpfa("flow").update_discretization(g, data)
    store["perm_new"] = np.ascontiguousarray(K_new.values)
    store["modified_cells"] = np.asarray(modified_cells, dtype=np.int64)
    for k in KEYS:
        pack_csr(f"upd_{k}", data[pp.DISCRETIZATION_MATRICES]["flow"][k], store)
    full = discretize(g, K_new, bc)  # the update must equal a fresh discretization
    for k in KEYS:
        d = abs(full[pp.DISCRETIZATION_MATRICES]["flow"][k] - data[pp.DISCRETIZATION_MATRICES]["flow"][k]).max()
        assert d < 1e-12, (k, d)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:32s} cells={g.num_cells:5d} {os.path.getsize(path)/1024:.0f} KiB")


def main():
    rng = np.random.default_rng(77)
    # the reference's own partial-discretization setting (tests/numerics/fv/test_mpfa.py:503-640)
    g = pp.CartGrid([5, 5]); g.compute_geometry()
    nc = g.num_cells
    kxx, kyy = 1 + rng.random(nc), 1 + rng.random(nc)
    K = pp.SecondOrderTensor(kxx=kxx, kyy=kyy, kxy=0.3 * rng.random(nc))
    bc = mixed_bc(g, ["dir", "neu"])
    cn = g.cell_nodes()
    specs = [
        {"specified_nodes": cn[:, 10].nonzero()[0]},   # cell at the domain boundary
        {"specified_nodes": cn[:, 12].nonzero()[0]},   # interior cell
        {"specified_cells": np.array([12])},
        {"specified_cells": np.array([0, 24])},
        {"specified_faces": np.array([14])},
    ]
    kxx2 = kxx.copy(); kxx2[[7, 12]] *= 10.0
    K2 = pp.SecondOrderTensor(kxx=kxx2, kyy=kyy, kxy=K.values[0, 1])
    save("partial_cart2d_5x5", g, K, bc, specs, K2, [7, 12])

    g = perturb_interior(pp.StructuredTetrahedralGrid([3, 3, 3], [1, 1, 1]), rng, 0.06)
    nc = g.num_cells
    k = 1 + rng.random(nc)
    K = pp.SecondOrderTensor(kxx=k, kyy=2 * k, kzz=0.5 * k, kxy=0.2 * k, kxz=0.05 * k, kyz=0.1 * k)
    bc = mixed_bc(g, ["dir", "neu", "rob"])
    cn = g.cell_nodes()
    specs = [
        {"specified_nodes": cn[:, 40].nonzero()[0]},
        {"specified_cells": np.array([5, 80])},
        {"specified_faces": np.array([100])},
    ]
    k2 = k.copy(); k2[[3, 50, 51]] *= 7.0
    K2 = pp.SecondOrderTensor(kxx=k2, kyy=2 * k2, kzz=0.5 * k2, kxy=0.2 * k2, kxz=0.05 * k2, kyz=0.1 * k2)
    save("partial_tet3d_3x3x3", g, K, bc, specs, K2, [3, 50, 51])


if __name__ == "__main__":
    main()
