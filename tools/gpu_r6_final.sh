#!/bin/bash
# Round 6 final: whole GPU suite, the default bench line, kernel trace of the bench, PMC passes.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6fin${1:-}
mkdir -p $O
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $O/timeline.log; }
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log; stamp pytest
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; stamp bench
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
try:
    d = json.loads([l for l in open(f"{o}/bench_default.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"bench: ms/step {d['ms_per_step']:.2f} cold {d['ms_per_step_cold']:.2f} value {d['value']:.3e} its {d['config']['iterations']} asm {d['assembly']['ms']:.2f} frac {d['assembly']['frac_of_hbm_peak']:.3f} {ph}")
    print(" roofline", d["roofline"]["name"], round(d["roofline"]["frac"], 3), {k["name"]: round(k["frac"], 3) for k in d["roofline_kernels"]})
    print(" cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None, "c2", d["config_c2"]["ms_per_step"], "c4", d["config_c4"]["ms_per_step"], d["config_c4"]["phases_ms"], "its", d["config_c4"]["iterations"])
    print(" whole grid", {k: d["whole_grid_check"].get(k) for k in ("pattern_row_lengths_equal", "flux_nnz_device", "flux_nnz_outside_neumann_rows_device", "rows_with_a_different_length_outside_neumann_rows")})
    print(" launches/it", d["launches_per_iteration"], "opapi", {k: round(v["ms"], 1) for k, v in d["operator_api"].items() if isinstance(v, dict) and "ms" in v})
except Exception as e:
    print("bench FAILED", e, open(f"{o}/bench_default.err").read()[-1500:])
PY
bash tools/gpu_trace_bench.sh > $O/trace_stdout.log 2>&1; cp gpurun_out/bench_kernel_stats.txt $O/bench_kernel_stats.txt; head -30 $O/bench_kernel_stats.txt | cut -c1-200; stamp trace
bash tools/gpu_pmc.sh > $O/pmc_stdout.log 2>&1
cp gpurun_out/pmc_summary.txt $O/pmc_summary.txt; cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json; cat $O/pmc_traffic.json | head -40; stamp pmc
rm -rf gpurun_out/btrace
# the randomized differential driver on the device against the reference itself (archive oracle/_ref)
ENVF=$(python - <<'PY'
import oracle
e = oracle.ref_env(extra_last=["."], prefer_archive=True)
print(e["PYTHONPATH"] if e else "")
PY
)
if [ -n "$ENVF" ] && [ -z "${SKIP_FUZZ:-}" ]; then
  (cd /tmp && PFV_FUZZ_DEVICE=1 PYTHONDONTWRITEBYTECODE=1 PYTHONPATH="$ENVF:$R" timeout 600 python $R/tools/fuzz_vs_reference.py 40 918273 > $R/$O/fuzz_device_vs_reference.log 2>&1; tail -2 $R/$O/fuzz_device_vs_reference.log)
  (cd /tmp && PFV_FUZZ_DEVICE=1 PYTHONDONTWRITEBYTECODE=1 PYTHONPATH="$ENVF:$R" timeout 600 python $R/tools/fuzz_vs_reference.py 30 555111 special > $R/$O/fuzz_device_vs_reference_special.log 2>&1; tail -2 $R/$O/fuzz_device_vs_reference_special.log)
  (cd /tmp && PFV_FUZZ_DEVICE=1 PYTHONDONTWRITEBYTECODE=1 PYTHONPATH="$ENVF:$R" timeout 600 python $R/tools/fuzz_vs_reference.py 40 7000 contrast > $R/$O/fuzz_device_vs_reference_contrast.log 2>&1; tail -2 $R/$O/fuzz_device_vs_reference_contrast.log)
  (cd /tmp && PFV_FUZZ_DECADES=2,6 PFV_FUZZ_DEVICE=1 PYTHONDONTWRITEBYTECODE=1 PYTHONPATH="$ENVF:$R" timeout 600 python $R/tools/fuzz_vs_reference.py 30 7100 contrast > $R/$O/fuzz_device_vs_reference_contrast_1e2_1e6.log 2>&1; tail -2 $R/$O/fuzz_device_vs_reference_contrast_1e2_1e6.log)
  (cd /tmp && PFV_FUZZ_DECADES=6,10 PFV_FUZZ_DEVICE=1 PYTHONDONTWRITEBYTECODE=1 PYTHONPATH="$ENVF:$R" timeout 900 python $R/tools/fuzz_vs_reference.py 60 9000 contrast > $R/$O/fuzz_device_contrast_1e6_1e10.log 2>&1; tail -2 $R/$O/fuzz_device_contrast_1e6_1e10.log)
  (cd /tmp && PFV_FUZZ_DECADES=10,15 PFV_FUZZ_DEVICE=1 PYTHONDONTWRITEBYTECODE=1 PYTHONPATH="$ENVF:$R" timeout 900 python $R/tools/fuzz_vs_reference.py 60 9100 contrast > $R/$O/fuzz_device_contrast_1e10_1e15.log 2>&1; tail -2 $R/$O/fuzz_device_contrast_1e10_1e15.log)
fi
stamp fuzz
# the command the driver runs at round end
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
try:
    d = json.loads([l for l in open(f"{o}/bench_driver_cmd.json") if l.startswith("{")][-1])
    print(f"driver cmd: ms/step {d['ms_per_step']:.2f} cold {d['ms_per_step_cold']:.2f} its {d['config']['iterations']} asm frac {d['assembly']['frac_of_hbm_peak']:.3f} roofline {d['roofline']['frac']:.3f}")
except Exception as e:
    print("driver-cmd bench FAILED", e, open(f"{o}/bench_driver_cmd.err").read()[-1500:])
PY
stamp bench20
cat $O/timeline.log
