"""Round-4 lab: the face kernel (and the interaction-region kernel) of the headline grid under environment
switches, one process, one grid upload.  Every variant is timed with HIP events (pfv_time_kernel) and its
flux / vector_source rows are compared bit for bit with the default variant's (the scheduling of the faces must not
change a single value: every entry adds its <= nodes(f) terms in node order).

usage: python tools/face_lab.py [n_side] [only <substring>]
With PFV_LAB_ONE="NAME" runs that one variant three times and exits (for rocprofv3 --pmc passes)."""
import hashlib
import os
import sys
import time

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

VARIANTS = {
    "base": {},
    "dyn": {"PFV_FACE_DYN": 1},
    "dyn_run2": {"PFV_FACE_DYN": 1, "PFV_FACE_RUN": 2},
    "dyn_run4": {"PFV_FACE_DYN": 1, "PFV_FACE_RUN": 4},
    "run2": {"PFV_FACE_RUN": 2},
    "cell": {"PFV_FACE_ORDER": 1},
    "cell_run2": {"PFV_FACE_ORDER": 1, "PFV_FACE_RUN": 2},
    "cell_run4": {"PFV_FACE_ORDER": 1, "PFV_FACE_RUN": 4},
    "cell_dyn": {"PFV_FACE_ORDER": 1, "PFV_FACE_DYN": 1},
    "cell_dyn_run2": {"PFV_FACE_ORDER": 1, "PFV_FACE_DYN": 1, "PFV_FACE_RUN": 2},
    "cell_dyn_run4": {"PFV_FACE_ORDER": 1, "PFV_FACE_DYN": 1, "PFV_FACE_RUN": 4},
    "cell_dyn_run8": {"PFV_FACE_ORDER": 1, "PFV_FACE_DYN": 1, "PFV_FACE_RUN": 8},
    "cell_dyn_run4_nt": {"PFV_FACE_ORDER": 1, "PFV_FACE_DYN": 1, "PFV_FACE_RUN": 4, "PFV_FACE_NT": 1},
}
KEYS = ("PFV_FACE_DYN", "PFV_FACE_RUN", "PFV_FACE_ORDER", "PFV_FACE_NT", "PFV_NODE_GJ")
one = os.environ.get("PFV_LAB_ONE")
only = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "only" else None

if os.environ.get("PFV_LAB_EMUL"):  # build container: checks the script and the ordering code on the host build
    from tests import _parity as P

    ctx = pa.Context(0, P.emulation_library())
else:
    ctx = pa.Context(0)
ctx.set_grid(raw)
ctx.set_params(Kvals, flags, None, eta)


def digest(which):
    rows = np.arange(0, ctx.matrix_info(which)[0], 97)
    m = ctx.matrix_rows(which, rows)
    return hashlib.sha1(m.data.tobytes()).hexdigest()[:12], float(abs(m.data).sum())


ref = None
for name, env in VARIANTS.items():
    if one and name != one:
        continue
    if only and only not in name:
        continue
    for k in KEYS:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    t0 = time.perf_counter()
    ctx.discretize(rebuild_topology=True)
    ctx.sync()
    t_disc = 1e3 * (time.perf_counter() - t0)
    ctx.discretize(rebuild_topology=True)
    st = ctx.stats()
    face = ctx.time_kernel(2, reps=3 if one else 5)
    d = (digest(0), digest(4))
    if ref is None:
        ref = d
    same = "bit-identical" if d == ref else f"DIFFERENT {d} vs {ref}"
    print(f"{name:18s} face {face:6.2f} ms   discretize {st['discretize_ms'] if 'discretize_ms' in st else t_disc:6.2f} "
          f"(topology {st['topology_ms']:.2f} symbolic {st['symbolic_ms']:.2f} node {st['node_ms']:.2f} "
          f"face-in-step {st['face_ms']:.2f})   {same}", flush=True)
if os.environ.get("PFV_LAB_NODE"):
    for k in KEYS:
        os.environ.pop(k, None)
    for gj in os.environ["PFV_LAB_NODE"].split(","):
        os.environ["PFV_NODE_GJ"] = gj
        ctx.discretize(rebuild_topology=False)
        node = ctx.time_kernel(1, reps=3)
        print(f"PFV_NODE_GJ={gj}: node kernel {node:.2f} ms   {digest(0)} {digest(4)}", flush=True)
