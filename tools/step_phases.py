import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, porepy_amd as pa
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.set_device(0); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
lp, Kv, fl, bv, src, eta = bench.make_slab_problem(69, 0, 1)
ctx = pa.Context(0); ctx.set_grid(lp.raw); ctx.set_params(Kv, fl, None, eta)
for it in range(4):
    ctx.discretize(rebuild_topology=True)
    st = ctx.stats(); print("step", it, "before solve: node %.1f face %.1f sym %.1f" % (st["node_ms"], st["face_ms"], st["symbolic_ms"]), flush=True)
    ctx.assemble(bv, None, src)
    if it >= 1:
        x, info = ctx.solve("bicgstab", rtol=1e-10, maxit=20000, raise_on_fail=False)
        print("   solve ms", info["solve_ms"], info["iterations"])
