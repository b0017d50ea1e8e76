"""Assembly-kernel lab (one process per variant; the variant is the environment): builds the headline
grid of bench.py (cached as .npz under /tmp on the box), discretizes, and times the interaction-region
(node) kernel and the face kernel with HIP events (pfv_time_kernel).  Prints one line.
  python tools/asm_lab.py LABEL [n_side]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import porepy_amd as pa  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "default"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 69
cache = f"/tmp/pfv_lab_{n}.npz"
if os.path.exists(cache):
    z = np.load(cache)
    raw = {k[4:]: z[k] for k in z.files if k.startswith("raw_")}
    raw["dim"] = int(z["dim"])
    raw["name"] = str(z["name"])
    Kvals, flags, bv, src = z["K"], z["flags"], z["bv"], z["src"]
    eta = float(z["eta"])
else:
    lp, Kvals, flags, bv, src, eta = bench.make_slab_problem(n, 0, 1)
    raw = lp.raw
    np.savez(cache, K=Kvals, flags=flags, bv=bv, src=src, eta=eta, dim=raw["dim"], name=raw["name"],
             **{"raw_" + k: v for k, v in raw.items() if isinstance(v, np.ndarray)})
ctx = pa.Context(0)
ctx.set_grid(raw)
ctx.set_params(Kvals, flags, None, eta)
t0 = time.perf_counter()
ctx.discretize(rebuild_topology=True)
ctx.sync()
t_first = time.perf_counter() - t0
ctx.discretize(rebuild_topology=True)
st = dict(ctx.stats())
node = min(ctx.time_kernel(1, 3) for _ in range(2))
face = min(ctx.time_kernel(2, 3) for _ in range(2))
t0 = time.perf_counter()
for _ in range(3):
    ctx.discretize(rebuild_topology=False)
ctx.sync()
warm = (time.perf_counter() - t0) / 3
ctx.assemble(bv, None, src)
x, info = ctx.solve("bicgstab", rtol=1e-10, maxit=2000, raise_on_fail=False, precond="amg")
A = ctx.matrix(pa._lib.MAT_SYSTEM)
res = float(np.linalg.norm(ctx.rhs() - A @ x) / np.linalg.norm(ctx.rhs()))
flux = ctx.matrix(pa._lib.MAT_FLUX)
print(json.dumps({"label": label, "node_ms": round(node, 3), "face_ms": round(face, 3),
                  "discretize_cold_ms": round(st["discretize_ms"], 2), "discretize_warm_ms": round(1e3 * warm, 2),
                  "topology_ms": round(st["topology_ms"], 2), "symbolic_ms": round(st["symbolic_ms"], 2),
                  "assemble_ms": round(ctx.stats()["assemble_ms"], 2),
                  "its": info["iterations"], "res": res, "flux_checksum": float(abs(flux.data).sum()),
                  "first_call_s": round(t_first, 2)}), flush=True)
