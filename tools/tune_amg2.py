"""AMG sweep 2 on the benchmark system: coarsest-level size, passes, alpha/omega with the f32 windowed cycle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import porepy_amd as pa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 69
g, K, bc, bv, src = bench.make_problem(n)
ctx = pa.Context(0)
ctx.set_grid(pa.grid_to_raw(g))
ctx.set_params(K.values, pa.bc_flags(bc), None, 1.0 / 3.0)
ctx.discretize(skip_vector_source=True)
combos = [(160, 3, 150, 80), (400, 3, 150, 80), (1000, 3, 150, 80), (400, 3, 140, 80), (400, 3, 160, 80), (400, 3, 150, 90),
          (400, 3, 150, 70), (400, 4, 150, 80), (400, 2, 130, 80)]
for target, passes, alpha, omega in combos:
    os.environ.update(PFV_AMG_COARSE_TARGET=str(target), PFV_AMG_PASSES=str(passes), PFV_AMG_ALPHA_PCT=str(alpha),
                      PFV_AMG_OMEGA_PCT=str(omega))
    best = None
    for rep in range(2):
        ctx.discretize(skip_vector_source=True)  # invalidates the system -> fresh setup
        ctx.assemble(bv, None, src)
        x, info = ctx.solve("bicgstab", rtol=1e-10, maxit=3000, raise_on_fail=False, precond="amg")
        best = info if best is None or info["solve_ms"] < best["solve_ms"] else best
    st = ctx.stats()
    print(f"target {target:4d} passes {passes} alpha {alpha/100:.2f} omega {omega/100:.2f}: its {best['iterations']:3d} solve {best['solve_ms']:6.1f} ms "
          f"(setup {st['amg_setup_ms']:5.1f}) levels {st['amg_levels']} cx {st['amg_operator_complexity']:.3f} coarsest {st['amg_coarsest_rows']}", flush=True)
