#!/bin/bash
# Round 6, checkpoint A: whole GPU suite, default bench line, the node || face overlap lab, the differential driver's
# contrast mode on the device (mechanics: the double-double regions).
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6a
mkdir -p $O
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $O/timeline.log; }
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log; stamp pytest
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; stamp bench
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
try:
    d = json.loads([l for l in open(f"{o}/bench_default.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"bench: ms/step {d['ms_per_step']:.2f} cold {d['ms_per_step_cold']:.2f} value {d['value']:.3e} its {d['config']['iterations']} asm {d['assembly']['ms']:.2f} frac {d['assembly']['frac_of_hbm_peak']:.3f} {ph}")
    print(" kept", d["config"]["pattern_reuse"]["csr_patterns_kept"], "roofline", d["roofline"]["name"], round(d["roofline"]["frac"], 3), {k["name"]: round(k["frac"], 3) for k in d["roofline_kernels"]})
    print(" c2", d["config_c2"]["ms_per_step"], "c4", d["config_c4"]["ms_per_step"], d["config_c4"]["phases_ms"], "its", d["config_c4"]["iterations"])
except Exception as e:
    print("bench FAILED", e, open(f"{o}/bench_default.err").read()[-1500:])
PY
for lab in 0 1; do
  PFV_LAB_NODE_FACE_OVERLAP=$lab timeout 600 python bench.py --steps 6 --warmup 3 --no-cold --no-cpu-baseline --no-whole-grid-check --no-extra-configs > $O/bench_lab$lab.json 2> $O/bench_lab$lab.err
  python - "$O" "$lab" <<'PY'
import json, sys
o, lab = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"{o}/bench_lab{lab}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"lab {lab}: ms/step {d['ms_per_step']:.2f} {ph}")
except Exception as e:
    print("lab FAILED", e)
PY
  grep "\[lab\]" $O/bench_lab$lab.err | tail -4
done
stamp lab
ENVF=$(python - <<'PY'
import oracle
e = oracle.ref_env(extra_last=["."], prefer_archive=True)
print(e["PYTHONPATH"] if e else "")
PY
)
if [ -n "$ENVF" ]; then
  (cd /tmp && PFV_FUZZ_DECADES=6,10 PFV_FUZZ_DEVICE=1 PYTHONDONTWRITEBYTECODE=1 PYTHONPATH="$ENVF:$R" timeout 900 python $R/tools/fuzz_vs_reference.py 30 9000 contrast > $R/$O/fuzz_device_contrast_1e6_1e10.log 2>&1; tail -2 $R/$O/fuzz_device_contrast_1e6_1e10.log)
  (cd /tmp && PFV_FUZZ_DECADES=10,15 PFV_FUZZ_DEVICE=1 PYTHONDONTWRITEBYTECODE=1 PYTHONPATH="$ENVF:$R" timeout 900 python $R/tools/fuzz_vs_reference.py 30 9100 contrast > $R/$O/fuzz_device_contrast_1e10_1e15.log 2>&1; tail -2 $R/$O/fuzz_device_contrast_1e10_1e15.log)
fi
stamp fuzz
cat $O/timeline.log
