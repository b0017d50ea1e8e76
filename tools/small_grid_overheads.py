"""Where the ~18 ms of a tiny grid's discretization go on the device (per phase, ms): handle creation, upload,
parameters, discretize, the six fetches, close."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import porepy_amd as pa  # noqa: E402
from porepy_amd import _lib  # noqa: E402
from porepy_amd.params import bc_flags  # noqa: E402

g = pa.CartGrid([5, 5], [1.0, 1.0])
g.compute_geometry()
raw = pa.grid_to_raw(g)
K = pa.SecondOrderTensor(kxx=np.ones(g.num_cells))
bc = pa.BoundaryCondition(g, g.get_all_boundary_faces()[:3], ["dir"] * 3)
acc = {}


def tick(name, t0):
    acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)


for rep in range(12):
    t = time.perf_counter(); ctx = _lib.Context(0); tick("create", t)
    t = time.perf_counter(); ctx.set_grid(raw); tick("set_grid", t)
    t = time.perf_counter(); ctx.set_params(np.asarray(K.values, float), bc_flags(bc), np.asarray(bc.robin_weight, float), 0.0, None); tick("set_params", t)
    t = time.perf_counter(); ctx.discretize(rebuild_topology=False); tick("discretize", t)
    t = time.perf_counter(); ctx.discretize(rebuild_topology=False); tick("discretize again", t)
    t = time.perf_counter(); ms = [ctx.matrix(w) for w in (_lib.MAT_FLUX, _lib.MAT_BOUND_FLUX, _lib.MAT_BOUND_PRESSURE_CELL, _lib.MAT_BOUND_PRESSURE_FACE, _lib.MAT_VECTOR_SOURCE, _lib.MAT_BOUND_PRESSURE_VECTOR_SOURCE)]; tick("six fetches", t)
    t = time.perf_counter(); ctx.close(); tick("close", t)
for k, v in acc.items():
    print(f"{k:18s} median {np.median(v[2:]):7.2f} ms   first {v[0]:7.2f}")
