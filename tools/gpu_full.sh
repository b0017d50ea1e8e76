#!/bin/bash
# One gpurun call: GPU parity suite + the default bench line (+ optional extra commands given as arguments).
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
mkdir -p $O
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $O/timeline.log; }
: > $O/timeline.log
timeout 900 python -m pytest tests -m gpu -q --timeout 400 --durations=8 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.log
stamp pytest
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
stamp bench_default
for c in "$@"; do
  bash -c "$c"
  stamp "$c"
done
tail -12 $O/pytest_gpu.log
tail -c 600 $O/bench_default.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "its", d["config"]["iterations"], "res", d["config"]["true_rel_residual"])
    print("field", d["config"]["field_error"])
    print("roofline", d["roofline"]["name"], d["roofline"]["frac"], {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
    print("triad", d["hbm_triad_measured_GBs"])
    print("assembly", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["assembly"].items() if k not in ("phases_note",)})
    print("opapi", d["operator_api"])
    print("cpu", d["cpu_baseline"])
    print("c2", d["config_c2"])
    print("c4", d["config_c4"])
except Exception as e:
    print("bench parse failed", e)
PY
cat $O/timeline.log
