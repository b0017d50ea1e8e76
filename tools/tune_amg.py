"""AMG parameter sweep on the benchmark system: passes per level, over-correction alpha, Jacobi omega."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import porepy_amd as pa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 69
g, K, bc, bv, src = bench.make_problem(n)
ctx = pa.Context(0)
ctx.set_grid(pa.grid_to_raw(g))
ctx.set_params(K.values, pa.bc_flags(bc), None, 1.0 / 3.0)
combos = [(3, 150, 70), (3, 130, 70), (3, 150, 80), (3, 130, 80), (3, 100, 70), (2, 100, 70), (2, 130, 70), (2, 130, 80), (4, 150, 70)]
for method in ("bicgstab", "gmres"):
    for passes, alpha, omega in combos if method == "bicgstab" else combos[:3]:
        os.environ.update(PFV_AMG_PASSES=str(passes), PFV_AMG_ALPHA_PCT=str(alpha), PFV_AMG_OMEGA_PCT=str(omega))
        ctx.discretize(skip_vector_source=True)
        ctx.assemble(bv, None, src)
        x, info = ctx.solve(method, rtol=1e-10, maxit=3000, raise_on_fail=False, precond="amg", restart=60)
        st = ctx.stats()
        print(f"{method:8s} passes {passes} alpha {alpha/100:.1f} omega {omega/100:.1f}: its {info['iterations']:4d} solve {info['solve_ms']:7.1f} ms "
              f"(setup {st['amg_setup_ms']:5.1f}) levels {st['amg_levels']} cx {st['amg_operator_complexity']:.3f} coarsest {st['amg_coarsest_rows']}", flush=True)
