"""Fracture planes one by one against one disjoint union (porepy_amd.Mpfa.discretize_batch): wall time of the
discretization of N small 2-D grids tilted in 3-D on the device, host arrays in, scipy matrices out.
usage: python tools/md_batch_bench.py [n_grids [cells_per_side]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import porepy_amd as pa  # noqa: E402

n_grids = int(sys.argv[1]) if len(sys.argv) > 1 else 52
side = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(0)


def plane(i):
    g = pa.CartGrid([side, side + i % 3], [1.0, 1.0])
    g.compute_geometry()
    a = 0.3 + 0.05 * i
    R = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    g.nodes = R @ g.nodes + rng.random(3)[:, None]
    g.compute_geometry()
    return g


def items():
    out = []
    for g in grids:
        sc = np.exp(0.3 * np.random.default_rng(g.num_cells).standard_normal(g.num_cells))
        K = pa.SecondOrderTensor(kxx=sc, kyy=2 * sc, kzz=sc, kxy=0.2 * sc)
        bf = g.get_all_boundary_faces()
        bc = pa.BoundaryCondition(g, bf[:3], ["dir"] * 3)
        out.append((g, pa.initialize_data({}, "flow", {"second_order_tensor": K, "bc": bc, "bc_values": np.zeros(g.num_faces),
                                                       "ambient_dimension": 3})))
    return out


grids = [plane(i) for i in range(n_grids)]
for label, fn in (("one by one", lambda d, it: [d.discretize(g, dat) for g, dat in it]),
                  ("one disjoint union", lambda d, it: d.discretize_batch(it))):
    best = 1e9
    for rep in range(3):
        d = pa.Mpfa("flow")
        it = items()
        t = time.perf_counter()
        fn(d, it)
        best = min(best, time.perf_counter() - t)
    print(f"{label:20s}: {n_grids} planes of {side} x {side}..{side + 2} cells: {best * 1e3:8.1f} ms  ({sum(g.num_cells for g in grids)} cells)")
