"""Where the time of the operator-API path goes on the headline grid (host side, perf_counter)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import porepy_amd as pa  # noqa: E402
from porepy_amd import mpfa as M  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 69
lp, Kvals, flags, bv, src, eta = bench.make_slab_problem(n, 0, 1)
g = pa.grid_from_raw(lp.raw)
K = type("K", (), {"values": Kvals})()
bc = type("BC", (), {"is_dir": (flags & 1) != 0, "is_neu": (flags & 2) != 0, "is_rob": np.zeros(flags.size, bool),
                     "is_internal": np.zeros(flags.size, bool), "robin_weight": np.ones(flags.size)})()
d = pa.Mpfa("flow", 0, lazy=True)
data = pa.initialize_data({}, "flow", {"second_order_tensor": K, "bc": bc, "bc_values": bv, "hip_rebuild_topology": True,
                                        "mpfa_eta": eta})
d.discretize(g, data)
d.assemble_matrix_rhs(g, data)
ctx = d.context(g)


def T(label, f, reps=2):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    print(f"{label:40s} {1e3 * (time.perf_counter() - t0) / reps:9.2f} ms", flush=True)
    return r


T("grid_fingerprint", lambda: M.grid_fingerprint(g))
T("estimate_device_bytes", lambda: M.estimate_device_bytes(g))
T("bc_flags", lambda: pa.bc_flags(bc))
T("set_params (K, flags, robin upload)", lambda: ctx.set_params(Kvals, flags, np.ones(flags.size), eta, None))
T("ctx.discretize(rebuild)", lambda: (ctx.discretize(rebuild_topology=True), ctx.sync()))
T("Mpfa.discretize lazy (whole)", lambda: d.discretize(g, data))
T("ctx.assemble", lambda: (ctx.assemble(bv, None, None), ctx.sync()))
T("ctx.matrix(A)", lambda: ctx.matrix(pa._lib.MAT_SYSTEM), reps=2)
os.environ["PFV_PINNED_COPIES"] = "0"
T("ctx.matrix(A) pageable copy", lambda: ctx.matrix(pa._lib.MAT_SYSTEM), reps=1)
os.environ["PFV_PINNED_COPIES"] = "1"
T("ctx.rhs()", lambda: ctx.rhs())
T("assemble_matrix_rhs (whole)", lambda: d.assemble_matrix_rhs(g, data))
T("np.empty+touch 1.6 GB", lambda: np.empty(137_000_000 * 12 // 8).fill(0.0), reps=1)
d.lazy = False
T("Mpfa.discretize eager (whole)", lambda: d.discretize(g, data), reps=1)
