"""One full step of the headline workload (cold discretize + div@flux + AMG-BiCGStab solve) without torch, for
rocprofv3 (kernel trace or --pmc passes): the grid of bench.py (cached .npz, see asm_lab.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import porepy_amd as pa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 69
cache = f"/tmp/pfv_lab_{n}.npz"
if os.path.exists(cache):
    z = np.load(cache)
    raw = {k[4:]: z[k] for k in z.files if k.startswith("raw_")}
    raw["dim"] = int(z["dim"])
    raw["name"] = str(z["name"])
    Kvals, flags, bv, src, eta = z["K"], z["flags"], z["bv"], z["src"], float(z["eta"])
else:
    lp, Kvals, flags, bv, src, eta = bench.make_slab_problem(n, 0, 1)
    raw = lp.raw
    np.savez(cache, K=Kvals, flags=flags, bv=bv, src=src, eta=eta, dim=raw["dim"], name=raw["name"],
             **{"raw_" + k: v for k, v in raw.items() if isinstance(v, np.ndarray)})
ctx = pa.Context(0)
ctx.set_grid(raw)
ctx.set_params(Kvals, flags, None, eta)
moving = os.environ.get("PFV_RUN_STEP_MOVING", "0") != "0"  # a new field for the second step (what bench.py times)
for it in range(2):
    if moving and it == 1:
        fac = np.exp(0.5 * np.random.default_rng(7).standard_normal(Kvals.shape[-1]))
        ctx.set_permeability(np.ascontiguousarray(Kvals * fac[None, None, :]))
    ctx.discretize(rebuild_topology=True)
    ctx.assemble(bv, None, src)
    x, info = ctx.solve("bicgstab", rtol=1e-13, maxit=2000, raise_on_fail=False, precond="amg")
print(info, {k: round(v, 2) for k, v in ctx.stats().items() if k.endswith("_ms")})
