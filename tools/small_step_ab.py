"""Small-system steps (what gates strong scaling): the cold step -- discretize + assemble + AMG setup + BiCGStab to 1e-13 --
on BASELINE configs[1] (196 608 tetrahedra, isotropic) and on one share of an 8-way split of the headline workload
(257 k cells of make_slab_problem(35)), under the environment switches given on the command line.
    python tools/small_step_ab.py name:"ENV=.. ENV2=.." ...   (runs every configuration in a subprocess)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    import numpy as np

    sys.path.insert(0, ROOT)
    import bench
    import porepy_amd as pa

    out = {}
    for name in ("c2_197k_isotropic", "share_257k_of_headline"):
        if name.startswith("c2"):
            g = pa.StructuredTetrahedralGrid([32] * 3, [1.0, 1.0, 1.0])
            g.compute_geometry()
            K = pa.SecondOrderTensor(np.ones(g.num_cells)).values
            bf = g.get_all_boundary_faces()
            flags = np.zeros(g.num_faces, dtype=np.uint8)
            flags[bf] = 1
            bv = np.zeros(g.num_faces)
            bv[bf] = g.face_centers[0, bf]
            raw, src, eta = pa.grid_to_raw(g), np.zeros(g.num_cells), 1.0 / 3.0
        else:
            lp, K, flags, bv, src, eta = bench.make_slab_problem(35, 0, 1)
            raw = lp.raw
        ctx = pa.Context(0)
        ctx.set_grid(raw)
        ctx.set_params(K, flags, None, eta)

        def step():
            ctx.discretize(rebuild_topology=True)
            ctx.assemble(bv, None, src)
            return ctx.solve("bicgstab", rtol=1e-13, maxit=5000, raise_on_fail=False, precond="amg")

        step()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            x, info = step()
        ctx.sync()
        dt = (time.perf_counter() - t0) / 5
        st = ctx.stats()
        out[name] = {"ms_per_step": round(1e3 * dt, 2), "its": info["iterations"], "res": info["rel_residual"],
                     "discretize": round(st["discretize_ms"], 2), "amg_setup": round(st["amg_setup_ms"], 2),
                     "solve": round(st["solve_ms"], 2), "levels": int(st["amg_levels"])}
        ctx.close()
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker()
    else:
        for v in sys.argv[1:]:
            n, _, e = v.partition(":")
            env = dict(os.environ)
            for kv in e.split():
                k, _, val = kv.partition("=")
                env[k] = val
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            print(f"{n:12s}", line[-1][7:] if line else "FAILED " + r.stderr[-400:], flush=True)
