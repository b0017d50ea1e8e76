import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, porepy_amd as pa
which = sys.argv[1]
if which == "cube":
    g, K, bc, bv, src = bench.make_problem(69)
    raw = pa.grid_to_raw(g); Kv = K.values; fl = pa.bc_flags(bc); eta = pa.determine_eta(g)
else:
    lp, Kv, fl, bv, src, eta = bench.make_slab_problem(69, 0, 1)
    raw = lp.raw
ctx = pa.Context(0); ctx.set_grid(raw); ctx.set_params(Kv, fl, None, eta)
ctx.discretize(rebuild_topology=True)
print(which, "face", min(ctx.time_kernel(2, 3) for _ in range(3)), "node", min(ctx.time_kernel(1, 3) for _ in range(2)), ctx.stats()["face_ms"])
