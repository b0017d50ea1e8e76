"""Assemble the benchmark system, build the AMG hierarchy (one solve) and launch the finest-level
smoothing product a few times (for PMC counter passes on k_amg_spmv)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, porepy_amd as pa
n = int(os.environ.get("TUNE_N", "69"))
g, K, bc, bv, src = bench.make_problem(n)
ctx = pa.Context(0); ctx.set_grid(pa.grid_to_raw(g))
ctx.set_params(K.values, pa.bc_flags(bc), bc.robin_weight, pa.determine_eta(g))
ctx.discretize(rebuild_topology=True)
ctx.assemble(bv, None, src)
x, info = ctx.solve(method="bicgstab", rtol=1e-10, maxit=200, precond="amg")
print(info, ctx.time_kernel(3, 5), ctx.time_kernel(0, 5))
