#!/bin/bash
# Round 5, call K: windows of the filtered operator derived from A's windows (PFV_WIN_DERIVE) A/B; small steps.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5k
mkdir -p $O
run() {  # tag, args, env...
  tag=$1; shift; args=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --no-whole-grid-check --no-cold $args > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
o, tag = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"{o}/bench_{tag}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"{tag}: ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} cells {d['config']['cells_per_gpu']} asm {d['assembly']['ms']:.2f} {ph} amg_setup {d['config']['amg']['setup_ms']:.2f} resid {d['config'].get('true_rel_residual')} transport {d['config'].get('transport')}")
except Exception as e:
    print(tag, "bench FAILED", e, open(f"{o}/bench_{tag}.err").read()[-1500:])
PY
}
run derive1 "--steps 8" PFV_WIN_DERIVE=1
run derive0 "--steps 8" PFV_WIN_DERIVE=0
run n35 "--steps 8 --n-side 35" PFV_X=0
run n35_sharded "--steps 8 --n-side 35 --force-sharded" PFV_X=0
run n44 "--steps 8 --n-side 44" PFV_X=0
run n55 "--steps 8 --n-side 55" PFV_X=0
run n69_sharded "--steps 5 --force-sharded" PFV_X=0
run n69_fixedk "--steps 8 --fixed-k" PFV_X=0
PFV_MPSA_X=0 timeout 600 python -m pytest tests/test_gpu_mpsa.py tests/test_gpu_parity.py -m gpu -q -x --timeout 500 -k "solve or robust or amg or c3 or uniaxial or mandel" > $O/pytest_solve.log 2>&1
echo "pytest exit $?" >> $O/pytest_solve.log; tail -3 $O/pytest_solve.log
