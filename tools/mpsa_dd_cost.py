"""What the double-double MPSA body costs (round 6): a tetrahedral box whose Lame parameters jump by 1e8 (a) across the plane
z = 0.5 (the nodes of one lattice layer are flagged), (b) from cell to cell at random (every node is flagged); mpsa_discretize
with the wide body (default) and with PFV_MPSA_DD=0 (FP64 body on every region).  GPU box:
    python tools/mpsa_dd_cost.py [n_side = 24]   -> profiles/r06_mpsa_dd_cost.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import porepy_amd as pa  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    g = pa.StructuredTetrahedralGrid([n, n, n], [1.0, 1.0, 1.0])
    g.compute_geometry()
    g = pa.perturb_interior_nodes(g, 0.2 / n)
    nc, nf = g.num_cells, g.num_faces
    bc = pa.BoundaryConditionVectorial(g)
    bf = g.get_all_boundary_faces()
    for axis in range(3):
        roll = bf[g.face_centers[axis, bf] < 1e-9]
        bc.is_dir[axis, roll] = True
        bc.is_neu[axis, roll] = False
    rng = np.random.default_rng(1)
    fields = {"homogeneous": np.ones(nc),
              "layer (1e8 across z = 0.5)": np.where(g.cell_centers[2] > 0.5, 1e8, 1.0),
              "random two-valued (1e8)": np.where(rng.random(nc) < 0.5, 1e8, 1.0)}
    print(f"# {nc} tetrahedra, {g.num_nodes} interaction regions (n = 108 in the interior); mpsa_discretize, ms (best of 3)")
    for name, s in fields.items():
        C = pa.FourthOrderTensor(s * np.ones(nc), s * np.ones(nc))
        row = []
        for dd in ("1", "0"):
            os.environ["PFV_MPSA_DD"] = dd
            ctx = pa.Context(0)
            ctx.set_grid(pa.grid_to_raw(g))
            ctx.mpsa_set_params(C.values, g.cell_volumes, bc.is_dir, bc.is_neu, 1.0 / 3.0)
            ctx.mpsa_discretize()
            ctx.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.mpsa_discretize()
                ctx.sync()
                best = min(best, 1e3 * (time.perf_counter() - t0))
            st = ctx.stats()
            row.append((best, int(st["mpsa_contrast_regions"]), float(st["mpsa_max_contrast"])))
            ctx.close()
        (t1, nreg, mx), (t0_, _, _) = row
        per = (t1 - t0_) / max(nreg, 1) * 1e3 if nreg else 0.0
        print(f"{name:32s} flagged {nreg:7d} (max contrast {mx:.1e})  wide body on {t1:9.2f}  off {t0_:8.2f}"
              + (f"  -> {per:.1f} us per flagged region" if nreg else ""))
    os.environ.pop("PFV_MPSA_DD", None)


if __name__ == "__main__":
    main()
