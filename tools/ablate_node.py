import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, porepy_amd as pa
n = int(os.environ.get("TUNE_N", "69"))
g, K, bc, bv, src = bench.make_problem(n)
ctx = pa.Context(0); ctx.set_grid(pa.grid_to_raw(g))
ctx.set_params(K.values, pa.bc_flags(bc), bc.robin_weight, pa.determine_eta(g))
ctx.discretize()
for ab in (0, 1, 2, 3, 4, 5, 6, 7):
    os.environ["PFV_NODE_ABLATE"] = str(ab)
    try:
        ms = min(ctx.time_kernel(1, 3) for _ in range(2))
    except Exception as e:
        ms = float("nan")
    print(f"ablate={ab} (1=noGJ 2=noFinish 4=noSetup): node kernel {ms:.2f} ms", flush=True)
os.environ["PFV_NODE_ABLATE"] = "0"
print("face kernel", min(ctx.time_kernel(2, 3) for _ in range(2)))
