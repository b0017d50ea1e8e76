"""W-cycle experiment on the benchmark system (PFV_AMG_GAMMA, PFV_AMG_GAMMA_LEVELS, alpha)."""
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
combos = [(1, 1, 150, 80), (2, 1, 150, 80), (2, 1, 160, 80), (2, 1, 170, 80), (2, 1, 150, 75), (2, 1, 150, 85), (2, 1, 160, 85), (2, 1, 140, 85)]
for gamma, glev, alpha, omega in combos:
    os.environ.update(PFV_AMG_GAMMA=str(gamma), PFV_AMG_GAMMA_LEVELS=str(glev), PFV_AMG_ALPHA_PCT=str(alpha), PFV_AMG_OMEGA_PCT=str(omega))
    best = None
    for rep in range(2):
        ctx.discretize(skip_vector_source=True)
        ctx.assemble(bv, None, src)
        x, info = ctx.solve("bicgstab", rtol=1e-10, maxit=3000, raise_on_fail=False, precond="amg")
        best = info if best is None or info["solve_ms"] < best["solve_ms"] else best
    st = ctx.stats()
    print(f"gamma {gamma} levels {glev} alpha {alpha/100:.2f} omega {omega/100:.2f}: its {best['iterations']:3d} solve {best['solve_ms']:6.1f} ms (setup {st['amg_setup_ms']:5.1f})", flush=True)
