#!/bin/bash
# Round 3 evidence call: GPU suite, PMC passes (-> pmc_traffic.json for this source hash), kernel trace, default bench.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd)
O=gpurun_out/$1
mkdir -p $O
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $O/timeline.log; }
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
timeout 1200 python -m pytest tests -m gpu -q --timeout 400 --durations=10 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log; stamp tests
# PMC passes (counters apart from the kernel trace)
bash tools/gpu_pmc.sh > $O/pmc_console.log 2>&1
cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null; cp gpurun_out/pmc_summary.txt $O/pmc_summary.txt 2>/dev/null
cp gpurun_out/pmc_traffic.json profiles/r03_pmc_traffic.json 2>/dev/null   # (so that the bench run below reports traffic)
stamp pmc
bash tools/gpu_r3_trace.sh $1 final: > $O/trace_console.log 2>&1
stamp trace
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
stamp bench_default
# device fuzz against PorePy itself (byte-compiled archive): plain cases, then the special ones (conditions per sub-face,
# partial discretization, tilted 2-D grids, TPFA, continuity points per sub-face)
(cd /tmp && PFV_FUZZ_DEVICE=1 PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=$R/oracle/shim:$R/oracle/_ref/porepy_ref.zip:$R timeout 400 python $R/tools/fuzz_vs_reference.py 40 11000 > $R/$O/fuzz_device_vs_reference.log 2>&1)
(cd /tmp && PFV_FUZZ_DEVICE=1 PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=$R/oracle/shim:$R/oracle/_ref/porepy_ref.zip:$R timeout 400 python $R/tools/fuzz_vs_reference.py 30 12000 special > $R/$O/fuzz_device_vs_reference_special.log 2>&1)
tail -n 2 $O/fuzz_device_vs_reference.log; tail -n 2 $O/fuzz_device_vs_reference_special.log
stamp fuzz
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
try:
    d = json.loads([l for l in open(f"{o}/bench_default.json") if l.startswith("{")][-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "its", d["config"]["iterations"], "res", d["config"]["true_rel_residual"])
    print("roofline", json.dumps(d["roofline"])[:600])
    print("kernels", [(k["name"], round(k["frac"], 3), round(k["ms_per_step"], 2), k.get("traffic")) for k in d["roofline_kernels"]])
    print("assembly", d["assembly"]["ms"], d["assembly"]["frac_of_hbm_peak"])
    print("operator_api", json.dumps(d["operator_api"])[:500])
    print("c2", d["config_c2"]["ms_per_step"], "c4", d["config_c4"]["ms_per_step"])
except Exception as e:
    print("bench parse failed", e, open(f"{o}/bench_default.err").read()[-1500:])
PY
cat $O/timeline.log
