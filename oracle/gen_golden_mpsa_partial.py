"""Golden fixtures for partial MPSA discretization / update_discretization, made by running the
REFERENCE (numerics/fv/mpsa.py:196-216, 383-487).

TEST INFRASTRUCTURE; build container only:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_mpsa_partial.py
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import OUT, pack_csr, perturb_interior  # noqa: E402
from oracle.ref_bridge import grid_to_raw  # noqa: E402

KEYS = ("stress", "bound_stress", "bound_displacement_cell", "bound_displacement_face")


def discretize(g, C, bc, extra=None):
    params = {"fourth_order_tensor": C, "bc": bc, "inverter": "python"}
    params.update(extra or {})
    data = pp.initialize_data({}, "mech", params)
    pp.Mpsa("mech").discretize(g, data)
    return data


def save(name, g, mu, lam, bc, specs, modified_cells):
    C = pp.FourthOrderTensor(mu, lam)
    store = {}
    for k, v in grid_to_raw(g).items():
        store["grid_" + k] = np.asarray(v)
    store["bc_is_dir"], store["bc_is_neu"] = bc.is_dir, bc.is_neu
    store["mu"], store["lam"] = mu, lam
    store["num_partial"] = np.array(len(specs))
    for i, spec in enumerate(specs):
        data = discretize(g, C, bc, spec)
        pd = data[pp.PARAMETERS]["mech"]
        for kind in ("cells", "faces", "nodes"):
            store[f"p{i}_spec_{kind}"] = np.asarray(spec.get("specified_" + kind, np.array([-1])), dtype=np.int64)
        store[f"p{i}_active_faces"] = np.asarray(pd["active_faces"], dtype=np.int64)
        for k in KEYS:
            pack_csr(f"p{i}_{k}", data[pp.DISCRETIZATION_MATRICES]["mech"][k], store)
    data = discretize(g, C, bc)
    mu2, lam2 = mu.copy(), lam.copy()
    mu2[modified_cells] *= 5.0
    lam2[modified_cells] *= 0.5
    C2 = pp.FourthOrderTensor(mu2, lam2)
    data[pp.PARAMETERS]["mech"]["fourth_order_tensor"] = C2
    data["update_discretization"] = {"modified_cells": np.asarray(modified_cells)}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pp.Mpsa("mech").update_discretization(g, data)
    full = discretize(g, C2, bc)
    for k in KEYS:
        d = abs(full[pp.DISCRETIZATION_MATRICES]["mech"][k] - data[pp.DISCRETIZATION_MATRICES]["mech"][k]).max()
        assert d < 1e-11, (k, d)
    store["mu_new"], store["lam_new"] = mu2, lam2
    store["modified_cells"] = np.asarray(modified_cells, dtype=np.int64)
    for k in ("stress", "bound_stress"):
        pack_csr(f"upd_{k}", data[pp.DISCRETIZATION_MATRICES]["mech"][k], store)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:32s} cells={g.num_cells:5d} {os.path.getsize(path)/1024:.0f} KiB")


def main():
    rng = np.random.default_rng(31)
    g = perturb_interior(pp.StructuredTriangleGrid([4, 4], [1, 1]), rng, 0.06)
    nc = g.num_cells
    bc = pp.BoundaryConditionVectorial(g)
    bf = g.get_all_boundary_faces()
    low = bf[g.face_centers[1, bf] < 1e-9]
    bc.is_dir[:, low] = True
    bc.is_neu[:, low] = False
    left = bf[g.face_centers[0, bf] < 1e-9]
    bc.is_dir[0, left] = True
    bc.is_neu[0, left] = False
    cn = g.cell_nodes()
    specs = [{"specified_nodes": cn[:, 13].nonzero()[0]}, {"specified_cells": np.array([3, 20])},
             {"specified_faces": np.array([17])}]
    save("mpsapartial_tri2d_4x4", g, 1 + rng.random(nc), 0.5 + rng.random(nc), bc, specs, [5, 6, 22])

    g = perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.05)
    nc = g.num_cells
    bc = pp.BoundaryConditionVectorial(g)
    bf = g.get_all_boundary_faces()
    low = bf[g.face_centers[2, bf] < 1e-9]
    bc.is_dir[:, low] = True
    bc.is_neu[:, low] = False
    specs = [{"specified_cells": np.array([7])}]
    save("mpsapartial_tet3d_2x2x2", g, 1 + rng.random(nc), 0.5 + rng.random(nc), bc, specs, [11, 30])


if __name__ == "__main__":
    main()
