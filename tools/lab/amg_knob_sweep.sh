#!/bin/bash
# Round 6 lab: the AMG shape / damping switches on the final library, headline grid (is the default still the optimum?)
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6lab3
mkdir -p $O
run() {
  local name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --no-whole-grid-check --no-cold --steps 6 --warmup 2 \
    > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    a = d["config"]["amg"]
    print(f"{sys.argv[2]:22s} step {d['ms_per_step']:.2f} ms  its {d['each_timed_step']['iterations']}  solve {d['assembly']['phases_ms']['solve_ms']:.2f}  setup {a['setup_ms']:.2f}  levels {a['levels']}  launches/it {d['launches_per_iteration']:.1f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run base PFV_LAB=0
run gamma1 PFV_AMG_GAMMA=1
run gamma_levels2 PFV_AMG_GAMMA_LEVELS=2
run alpha140 PFV_AMG_ALPHA_PCT=140
run alpha160 PFV_AMG_ALPHA_PCT=160
run alpha170 PFV_AMG_ALPHA_PCT=170
run filter20 PFV_AMG_FILTER_PERMIL=20
run filter40 PFV_AMG_FILTER_PERMIL=40
run coarse80 PFV_AMG_COARSE_TARGET=80
run coarse300 PFV_AMG_COARSE_TARGET=300
run rho170 PFV_AMG_OMEGA_RHO_PCT=170
run rho190 PFV_AMG_OMEGA_RHO_PCT=190
run base_again PFV_LAB=0
