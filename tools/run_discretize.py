"""Run only the assembly phases (for profiling): discretize twice + assemble + a few SpMVs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, porepy_amd as pa
n = int(os.environ.get("TUNE_N", "69"))
g, K, bc, bv, src = bench.make_problem(n)
ctx = pa.Context(0); ctx.set_grid(pa.grid_to_raw(g))
ctx.set_params(K.values, pa.bc_flags(bc), bc.robin_weight, pa.determine_eta(g))
ctx.discretize(rebuild_topology=True)
ctx.discretize(rebuild_topology=True)
ctx.assemble(bv, None, src)
print(ctx.time_kernel(0, 5), ctx.stats())
