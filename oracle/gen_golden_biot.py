"""Golden fixtures for the Biot coupling terms (tests/golden/biot_*.npz) made by running the
REFERENCE pp.Biot (numerics/fv/biot.py:247-1135).

TEST INFRASTRUCTURE; build container only:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_biot.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import OUT, pack_csr, perturb_interior  # noqa: E402
from oracle.gen_golden_mpsa_robin import robin_bc  # noqa: E402
from oracle.ref_bridge import grid_to_raw  # noqa: E402

MECH = ("stress", "bound_stress", "bound_displacement_cell", "bound_displacement_face")
COUP = ("scalar_gradient", "displacement_divergence", "boundary_displacement_divergence", "mpsa_consistency",
        "bound_displacement_pressure")


def save(name, g, C, bc, alphas, keys=COUP, more_params=None, extra=None):
    params = {"fourth_order_tensor": C, "bc": bc, "inverter": "python", "scalar_vector_mappings": alphas}
    if more_params:
        params.update(more_params)
    data = pp.initialize_data({}, "mechanics", params)
    pp.Biot("mechanics").discretize(g, data)
    mats = data[pp.DISCRETIZATION_MATRICES]["mechanics"]
    store = {}
    for k, v in grid_to_raw(g).items():
        store["grid_" + k] = np.asarray(v)
    store["bc_is_dir"], store["bc_is_neu"] = np.asarray(bc.is_dir, bool), np.asarray(bc.is_neu, bool)
    store["bc_is_rob"], store["bc_robin_weight"] = np.asarray(bc.is_rob, bool), np.asarray(bc.robin_weight)
    store["stiffness"] = np.ascontiguousarray(C.values)
    store["alpha_keys"] = np.array(list(alphas.keys()))
    for key, al in alphas.items():
        v = al.values if hasattr(al, "values") else pp.SecondOrderTensor(al * np.ones(g.num_cells)).values
        store[f"alpha_{key}"] = np.ascontiguousarray(v)
        for k in keys:
            pack_csr(f"ref_{k}__{key}", mats[k][key], store)
    for k in ("stress", "bound_stress"):
        pack_csr("ref_" + k, mats[k], store)
    if extra:
        store.update(extra)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:34s} cells={g.num_cells:4d}  {os.path.getsize(path)/1024:.0f} KiB")


def main():
    rng = np.random.default_rng(808)
    g = perturb_interior(pp.StructuredTriangleGrid([3, 3], [1, 1]), rng, 0.08); nc = g.num_cells
    C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
    bc = robin_bc(g, rng, "mixed")
    a2 = pp.SecondOrderTensor(kxx=1 + rng.random(nc), kyy=0.5 + rng.random(nc), kxy=0.3 * rng.random(nc))
    save("biot_tri2d_3x3_mixed", g, C, bc, {"pressure": 0.8, "temperature": a2})
    g = pp.CartGrid([3, 2]); g.compute_geometry(); nc = g.num_cells
    C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
    bc = pp.BoundaryConditionVectorial(g, g.get_all_boundary_faces(), "dir")
    save("biot_cart2d_3x2_dir", g, C, bc, {"pressure": 1.0})
    g = perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.08); nc = g.num_cells
    C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
    bc = robin_bc(g, rng, "mixed")
    a3 = pp.SecondOrderTensor(kxx=1 + rng.random(nc), kyy=0.5 + rng.random(nc), kzz=0.7 + rng.random(nc),
                              kxy=0.2 * rng.random(nc), kxz=0.1 * rng.random(nc), kyz=0.15 * rng.random(nc))
    save("biot_tet_2x2x2_mixed", g, C, bc, {"pressure": a3})


if __name__ == "__main__":
    main()
