#!/bin/bash
# Round 5, late: parameter sweep of the AMG settings on the moving-K protocol (the defaults were tuned on a fixed K)
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5sweep
mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs --no-whole-grid-check --no-cold > $O/$tag.json 2> $O/$tag.err
  python - "$O/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:28s} ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} solve {d['assembly']['phases_ms']['solve_ms']:.2f} amg_setup {d['config']['amg']['setup_ms']:.2f} opc {d['config']['amg']['operator_complexity']:.3f} levels {d['config']['amg']['levels']} each {[round(x,1) for x in d.get('each_timed_step',{}).get('ms',[])] if isinstance(d.get('each_timed_step'),dict) else ''}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run default PFV_X=0
run filter20 PFV_AMG_FILTER_PERMIL=20
run filter40 PFV_AMG_FILTER_PERMIL=40
run filter50 PFV_AMG_FILTER_PERMIL=50
run power5 PFV_AMG_OMEGA_POWER_STEPS=5
run gamma1 PFV_AMG_GAMMA=1
run gammalev2 PFV_AMG_GAMMA_LEVELS=2
run alpha130 PFV_AMG_ALPHA_PCT=130
run alpha170 PFV_AMG_ALPHA_PCT=170
run rho170 PFV_AMG_OMEGA_RHO_PCT=170
run rho190 PFV_AMG_OMEGA_RHO_PCT=190
run default2 PFV_X=0
