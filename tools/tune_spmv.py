"""GPU tuning helper: time SpMV variants (selected through env hooks read per launch) on the
bench problem, one process, one grid."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, porepy_amd as pa

n = int(os.environ.get("TUNE_N", "69"))
g, K, bc, bv, src = bench.make_problem(n)
ctx = pa.Context(0); ctx.set_grid(pa.grid_to_raw(g))
ctx.set_params(K.values, pa.bc_flags(bc), bc.robin_weight, pa.determine_eta(g))
ctx.discretize(); ctx.assemble(bv, None, src)
nnz = ctx.matrix_info(6)[2]; nc = g.num_cells
by = 12.0 * nnz + 4.0 * (nc + 1) + 16.0 * nc
variants = [("L16U1", {"PFV_SPMV_L": "16", "PFV_SPMV_U": "1"}), ("L16U2", {"PFV_SPMV_L": "16", "PFV_SPMV_U": "2"}),
            ("L16U3", {"PFV_SPMV_L": "16", "PFV_SPMV_U": "3"}), ("L16U4", {"PFV_SPMV_L": "16", "PFV_SPMV_U": "4"}),
            ("L16U5", {"PFV_SPMV_L": "16", "PFV_SPMV_U": "5"}), ("L8U4", {"PFV_SPMV_L": "8", "PFV_SPMV_U": "4"}),
            ("L8U5", {"PFV_SPMV_L": "8", "PFV_SPMV_U": "5"}), ("L32U3", {"PFV_SPMV_L": "32", "PFV_SPMV_U": "3"}),
            ("L4U5", {"PFV_SPMV_L": "4", "PFV_SPMV_U": "5"}), ("L16U5nt", {"PFV_SPMV_L": "16", "PFV_SPMV_U": "5", "PFV_SPMV_NT": "1"}),
            ("default", {})]
for tag, env in variants:
    for k in ("PFV_SPMV_L", "PFV_SPMV_NT", "PFV_SPMV_BLOCKS", "PFV_SPMV_U"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ms = min(ctx.time_kernel(0, 30) for _ in range(3))
    print(f"{tag:10s} {ms*1e3:8.1f} us  {by/ms/1e6:8.1f} GB/s", flush=True)
print(json.dumps(ctx.stats()))
