"""BiCGStab vs GMRES(m) on the benchmark system (Jacobi preconditioner, rtol 1e-10)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import porepy_amd as pa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 69
g, K, bc, bv, src = bench.make_problem(n)
ctx = pa.Context(0)
ctx.set_grid(pa.grid_to_raw(g))
ctx.set_params(K.values, pa.bc_flags(bc), None, 1.0 / 3.0)
ctx.discretize()
ctx.assemble(bv, None, src)
for method, restart in (("bicgstab", 0), ("gmres", 30), ("gmres", 100)):
    x, info = ctx.solve(method, rtol=1e-10, maxit=20000, restart=restart, raise_on_fail=False)
    print(f"{method:9s} restart {restart:3d}: {info}", flush=True)
