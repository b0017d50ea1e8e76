"""Interaction-region kernel alone on the headline grid: time per launch (pfv_time_kernel 1) under switches.
usage: python tools/node_lab.py [n_side] -- prints one line per variant."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import porepy_amd as pa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 69
lp, Kvals, flags, bv, src, eta = bench.make_slab_problem(n, 0, 1)
ctx = pa.Context(0)
ctx.set_grid(lp.raw)
ctx.set_params(Kvals, flags, None, eta)
ctx.discretize(rebuild_topology=True)
variants = [("gj3", {"PFV_NODE_GJ": "3"}), ("gj5", {"PFV_NODE_GJ": "5"}),
            ("gj3 no GJ", {"PFV_NODE_GJ": "3", "PFV_NODE_ABLATE": "1"}),
            ("gj3 no finish", {"PFV_NODE_GJ": "3", "PFV_NODE_ABLATE": "2"}),
            ("gj3 setup only", {"PFV_NODE_GJ": "3", "PFV_NODE_ABLATE": "3"}),
            ("gj5 no finish", {"PFV_NODE_GJ": "5", "PFV_NODE_ABLATE": "2"})]
extra = os.environ.get("PFV_LAB_VARIANTS", "")
for item in extra.split(";"):
    if item.strip():
        name, _, kv = item.partition(":")
        variants.append((name.strip(), dict(p.split("=") for p in kv.split(","))))
for name, env in variants:
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ms = ctx.time_kernel(1, reps=5)
    except Exception as e:  # noqa: BLE001
        ms = float("nan")
        print(name, "failed:", e)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    print(f"{name:28s} {ms:8.3f} ms", flush=True)
