"""BASELINE configs[3] (MPSA, 511 104 tetrahedra) step under environment switches, one subprocess each.
    python tools/mpsa_step_ab.py name:"ENV=.. ENV2=.." ..."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    import numpy as np

    sys.path.insert(0, ROOT)
    import porepy_amd as pa

    n = int(os.environ.get("MPSA_N", "44"))
    g = pa.StructuredTetrahedralGrid([n, n, n], [1, 1, 1])
    g.compute_geometry()
    g = pa.perturb_interior_nodes(g, 0.2 / n)
    nc, nf = g.num_cells, g.num_faces
    C = pa.FourthOrderTensor(np.ones(nc), np.ones(nc))
    bc = pa.BoundaryConditionVectorial(g)
    bf = g.get_all_boundary_faces()
    fc = g.face_centers
    for axis in range(3):
        roll = bf[fc[axis, bf] < 1e-9]
        bc.is_dir[axis, roll] = True
        bc.is_neu[axis, roll] = False
    bv = np.zeros((3, nf))
    top = bf[fc[2, bf] > 1 - 1e-9]
    bv[2, top] = -g.face_areas[top]
    ctx = pa.Context(0)
    ctx.set_grid(pa.grid_to_raw(g))
    ctx.mpsa_set_params(C.values, g.cell_volumes, bc.is_dir, bc.is_neu, 1.0 / 3.0)

    def step():
        ctx.mpsa_discretize(rebuild_topology=True)
        ctx.mpsa_assemble(bv.ravel("F"), None)
        return ctx.solve("bicgstab", rtol=1e-13, maxit=5000, n=3 * nc, raise_on_fail=False, precond="amg")

    step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(2):
        u, info = step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / 2
    st = ctx.stats()
    u = u.reshape(3, -1, order="F")
    cc = g.cell_centers
    err = float(np.max(np.abs(u - np.vstack((0.1 * cc[0], 0.1 * cc[1], -0.4 * cc[2])))))
    print("RESULT " + json.dumps({"ms_per_step": round(1e3 * dt, 1), "its": info["iterations"], "res": info["rel_residual"],
                                  "err": err, "node": round(st["node_ms"], 1), "face": round(st["face_ms"], 1),
                                  "amg_setup": round(st["amg_setup_ms"], 1), "solve": round(st["solve_ms"], 1),
                                  "levels": int(st["amg_levels"]), "opc": round(st["amg_operator_complexity"], 3)}))


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
