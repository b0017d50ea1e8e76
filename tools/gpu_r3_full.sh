#!/bin/bash
# Round 3 full call: GPU suite, the default bench line (reference cpu_baseline, configs[1], configs[3], operator API),
# a kernel trace.  Outputs under gpurun_out/$1.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$1
mkdir -p $O
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $O/timeline.log; }
timeout 1200 python -m pytest tests -m gpu -q --timeout 400 --durations=10 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log; stamp tests
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
stamp bench_default
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
try:
    d = json.loads([l for l in open(f"{o}/bench_default.json") if l.startswith("{")][-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "its", d["config"]["iterations"], "res", d["config"]["true_rel_residual"])
    print("roofline", d["roofline"]["name"], d["roofline"]["frac"], [(k["name"], round(k["frac"], 3), round(k["ms_per_step"], 2)) for k in d["roofline_kernels"]])
    print("assembly", d["assembly"]["ms"], d["assembly"]["frac_of_hbm_peak"], d["assembly"]["phases_ms"])
    print("values_only_step", d["assembly"].get("values_only_step", {}).get("ms_per_step"))
    print("operator_api", json.dumps(d["operator_api"])[:900])
    print("cpu_baseline", json.dumps(d["cpu_baseline"])[:600])
    print("c2", json.dumps(d["config_c2"])[:500])
    print("c4", json.dumps(d["config_c4"])[:700])
except Exception as e:
    print("bench parse failed", e, open(f"{o}/bench_default.err").read()[-1500:])
PY
cat $O/timeline.log
