"""CPU lab (host-emulation build, TEST INFRASTRUCTURE): iteration counts of AMG-BiCGStab on the headline workload
family for sweeps of PFV_AMG_* environment switches -- algorithmic choices are made here, timings on the GPU.
    python tools/amg_cpu_lab.py N_SIDE VAR=v1,v2,... [VAR2=...]"""
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import porepy_amd as pa  # noqa: E402
from tests import _parity as P  # noqa: E402

n = int(sys.argv[1])
sweeps = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[2:]]
lp, Kvals, flags, bv, src, eta = bench.make_slab_problem(n, 0, 1)
ctx = pa.Context(0, P.emulation_library())
ctx.set_grid(lp.raw)
ctx.set_params(Kvals, flags, None, eta)
ctx.discretize(rebuild_topology=True, skip_vector_source=True)
ctx.assemble(bv, None, src)
print(f"{lp.raw['cell_centers'].shape[1]} cells")
for combo in itertools.product(*[v for _, v in sweeps]):
    for (k, _), v in zip(sweeps, combo):
        os.environ[k] = v
    os.environ["PFV_AMG_REUSE"] = "0"
    ctx.discretize(rebuild_topology=False, skip_vector_source=True)  # a new system: the hierarchy is rebuilt under the new switches
    ctx.assemble(bv, None, src)
    t0 = time.perf_counter()
    x, info = ctx.solve("bicgstab", rtol=1e-13, maxit=400, raise_on_fail=False, precond="amg")
    st = ctx.stats()
    print(" ".join(f"{k}={v}" for (k, _), v in zip(sweeps, combo)), "-> iterations", info["iterations"], "converged",
          info["converged"], f"res {info['rel_residual']:.1e} levels {int(st['amg_levels'])} opc {st['amg_operator_complexity']:.3f}"
          f" ({time.perf_counter() - t0:.1f} s)", flush=True)
