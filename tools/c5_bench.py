"""BASELINE configs[4] stand-in at a size that gives the device work: the reference's thermo-hydro
model (models/mass_and_energy_balance.py:83) on a 3-D box with 52 fractures, C5_N_SIDE lattice cells a side (32: 36 657
cells / 98 229 unknowns; 48: ~110 k 3-D cells), one time step.

    python tools/c5_bench.py [--reference]      (C5_N_SIDE, C5_MAX_EXTENT from the environment; defaults 32 / 14)
Runs tests/_dropin_c5_script.py's model in a subprocess with the reference importable (oracle.ref_env: the live tree or
the byte-compiled archive) with pp.Mpfa rebound to the PRODUCT library, every subdomain discretization on the device
(planes as disjoint unions) and every Newton system solved by the device's GMRES + block preconditioner; reports where
the wall time of the step goes (the operator trees are assembled by the reference's own AD code on the host -- not
part of the hot path).  --reference also runs the untouched reference (its Mpfa + scipy direct solves): minutes.
Prints one line RESULT {json}."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import json, os, sys, time
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("c5", os.path.join(sys.argv[1], "tests", "_dropin_c5_script.py"))
c5 = importlib.util.module_from_spec(spec); spec.loader.exec_module(c5)
import porepy as pp
from porepy_amd import md_sharding
mode = sys.argv[2]
T = {"discretize": 0.0, "assemble": 0.0, "solve": 0.0, "n_assemble": 0, "n_solve": 0, "n_discretize": 0}
def timed(cls, name, key, cnt):
    orig = getattr(cls, name)
    def f(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return orig(self, *a, **k)
        finally:
            T[key] += time.perf_counter() - t0; T[cnt] += 1
    setattr(cls, name, f)
base = c5.Model if mode == "reference" else c5.HipSolveModel
timed(base, "discretize", "discretize", "n_discretize")
timed(base, "assemble_linear_system", "assemble", "n_assemble")
timed(base, "solve_linear_system", "solve", "n_solve")
t0 = time.perf_counter()
if mode == "reference":
    out = c5.run()
    extra = {}
else:
    calls = c5.rebind()
    stats = {}
    with md_sharding.batched_discretization(pp, stats=stats):
        out = c5.run(c5.HipSolveModel, "hip_gmres", {"precond": "block", "rtol": 1e-12, "restart": 80})
    extra = {"device_calls": calls, "gmres_iterations": [s["iterations"] for s in out["solves"]],
             "worst_true_residual": max(s["true_rel_residual"] for s in out["solves"]),
             "library": str(c5.P.dropin_library()._name)}
wall = time.perf_counter() - t0
res = {"mode": mode, "wall_s": wall, "cells": out["cells"], "dofs": int(out["x"].size), "subdomains": out["dims"],
       "interfaces": out["n_intf"], "seconds": {k: v for k, v in T.items() if not k.startswith("n_")},
       "calls": {k: v for k, v in T.items() if k.startswith("n_")},
       "x_norm": float(np.linalg.norm(out["x"])), "T_range": [float(out["T"].min()), float(out["T"].max())]}
res.update(extra)
np.save(os.path.join(sys.argv[3], "x_" + mode + ".npy"), out["x"])
print("RESULT " + json.dumps(res), flush=True)
'''


def run(mode: str, tmp: str, timeout: float = 7200.0):
    import oracle

    env = oracle.ref_env(extra_first=[ROOT])
    if env is None:
        return {"error": "no reference importable"}
    env.setdefault("C5_N_SIDE", "32")
    env.setdefault("C5_MAX_EXTENT", "14")
    env["PFV_DROPIN_LIBRARY"] = os.environ.get("PFV_DROPIN_LIBRARY", "product")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, mode, tmp], env=env, cwd="/tmp", capture_output=True, text=True,
                       timeout=timeout)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        return {"error": (r.stderr or r.stdout)[-1500:]}
    return json.loads(line[-1][7:])


def main():
    import tempfile

    import numpy as np

    with tempfile.TemporaryDirectory(prefix="pfv_c5_") as tmp:
        out = {"n_side": int(os.environ.get("C5_N_SIDE", "32")), "device": run("device", tmp)}
        if "--reference" in sys.argv:
            out["reference"] = run("reference", tmp)
            try:
                a, b = np.load(os.path.join(tmp, "x_device.npy")), np.load(os.path.join(tmp, "x_reference.npy"))
                out["x_rel_diff_device_vs_reference"] = float(np.linalg.norm(a - b) / np.linalg.norm(b))
            except Exception as e:  # noqa: BLE001
                out["x_rel_diff_device_vs_reference"] = repr(e)
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
