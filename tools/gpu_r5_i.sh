#!/bin/bash
# Round 5, call I: LDS cache of raw table rows in the face kernel (PFV_FACE_CACHE = slots; cell-major face order).
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5i
mkdir -p $O
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --no-whole-grid-check --no-cold --steps 6 > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
o, tag = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"{o}/bench_{tag}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    fk = [k for k in [d["roofline"]] + d["roofline_kernels"] if k["name"] == "face_kernel"][0]
    print(f"{tag}: ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} asm {d['assembly']['ms']:.2f} {ph} face alone {fk['ms_per_launch']:.2f} resid {d['config']['true_rel_residual']:.1e}")
except Exception as e:
    print(tag, "bench FAILED", e, open(f"{o}/bench_{tag}.err").read()[-1500:])
PY
}
run cache0 PFV_FACE_CACHE=0
run order1 PFV_FACE_ORDER=1
run cache4 PFV_FACE_CACHE=4
run cache6 PFV_FACE_CACHE=6
run cache4_run8 PFV_FACE_CACHE=4 PFV_FACE_RUN=8
PFV_FACE_CACHE=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 800 -k "golden_case or timed_bench_grid or config_c2" > $O/pytest_cache.log 2>&1
echo "pytest exit $?" >> $O/pytest_cache.log; tail -4 $O/pytest_cache.log
