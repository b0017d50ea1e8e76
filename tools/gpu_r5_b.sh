#!/bin/bash
# Round 5, call B: the unpivoted, verified elimination (PFV_NODE_GJ=5) and the low-priority stream of the
# interaction-region kernel (PFV_NODE_LOWPRIO=1): parity under the switch, A/B of the step.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5b
mkdir -p $O
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $O/timeline.log; }
PFV_NODE_GJ=5 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 800 -k "golden_case or generic_pattern or timed_bench_grid or known_answers or operator_api or heterogeneous" > $O/pytest_gj5.log 2>&1
echo "pytest exit $?" >> $O/pytest_gj5.log; tail -4 $O/pytest_gj5.log; stamp tests
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --no-whole-grid-check --no-cold --steps 6 > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
o, tag = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"{o}/bench_{tag}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    nk = [k for k in [d["roofline"]] + d["roofline_kernels"] if k["name"] == "node_kernel"][0]
    print(f"{tag}: ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} asm {d['assembly']['ms']:.2f} {ph} node alone {nk['ms_per_launch']:.2f} resid {d['config']['true_rel_residual']:.1e}")
except Exception as e:
    print(tag, "bench FAILED", e, open(f"{o}/bench_{tag}.err").read()[-1500:])
PY
  stamp $tag
}
run base PFV_NODE_GJ=3
run gj5 PFV_NODE_GJ=5
run lowprio PFV_NODE_GJ=3 PFV_NODE_LOWPRIO=1
run gj5_lowprio PFV_NODE_GJ=5 PFV_NODE_LOWPRIO=1
python - <<'PY' > $O/redo.log 2>&1
import os, sys
sys.path.insert(0, ".")
os.environ["PFV_NODE_GJ"] = "5"
import numpy as np, bench, porepy_amd as pa
lp, K, fl, bv, src, eta = bench.make_slab_problem(69, 0, 1)
ctx = pa.Context(0); ctx.set_grid(lp.raw); ctx.set_params(K, fl, None, eta)
ctx.discretize(rebuild_topology=True); st = ctx.stats()
print("n_side 69: nodes", st["num_nodes"], "redone by the pivoted body", st["node_redo"], "node_ms", st["node_ms"])
for n, kind in ((40, "cart"),):
    g = pa.CartGrid([n, n, n], [1.0, 1.0, 1.0]); g.compute_geometry()
    Kc = pa.SecondOrderTensor(np.ones(g.num_cells))
    bf = g.get_all_boundary_faces(); bc = pa.BoundaryCondition(g, bf, ["dir"] * bf.size)
    c2 = pa.Context(0); c2.set_grid(pa.grid_to_raw(g)); c2.set_params(Kc.values, pa.bc_flags(bc), None, 0.0)
    c2.discretize(rebuild_topology=True); s2 = c2.stats()
    print(kind, n, "nodes", s2["num_nodes"], "redo", s2["node_redo"])
PY
cat $O/redo.log; stamp redo
cat $O/timeline.log
