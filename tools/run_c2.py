"""configs[1] (196 608 tetrahedra) steps without torch, for rocprofv3: is the small-system solve bound by the kernels
or by the rate the host can launch them?  Prints the wall time of the solves; compare with the kernel-time sum."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import porepy_amd as pa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = pa.StructuredTetrahedralGrid([n, n, n], [1.0, 1.0, 1.0])
g.compute_geometry()
K = pa.SecondOrderTensor(np.ones(g.num_cells))
bf = g.get_all_boundary_faces()
bc = pa.BoundaryCondition(g, bf, ["dir"] * bf.size)
bv = np.zeros(g.num_faces)
bv[bf] = g.face_centers[0, bf]
ctx = pa.Context(0)
ctx.set_grid(pa.grid_to_raw(g))
ctx.set_params(K.values, pa.bc_flags(bc), None, 1.0 / 3.0)
ctx.discretize(rebuild_topology=True)
ctx.assemble(bv, None, np.zeros(g.num_cells))
x, info = ctx.solve("bicgstab", rtol=1e-13, maxit=2000, raise_on_fail=False, precond="amg")
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    x, info = ctx.solve("bicgstab", rtol=1e-13, maxit=2000, raise_on_fail=False, precond="amg")
ctx.sync()
dt = (time.perf_counter() - t0) / reps
print(f"cells {g.num_cells}: solve wall {1e3 * dt:.2f} ms, {info['iterations']} iterations, library solve_ms {ctx.stats()['solve_ms']:.2f} "
      f"(hierarchy kept between these solves), {reps + 1} solves in the trace")
