#!/bin/bash
# Round 6 lab: the fused forms of the AMG cycle's larger coarse levels (PFV_AMG_FUSE_CYCLE) against the launches they replace
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6lab2
mkdir -p $O
run() {  # name, bench flags..., env via env
  local name=$1; shift
  timeout 900 python bench.py --no-cpu-baseline --no-whole-grid-check --no-cold --steps 6 --warmup 2 "$@" \
    > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    s = f"{sys.argv[2]:12s} step {d['ms_per_step']:.2f} ms  its {d['config']['iterations']}  solve {d['assembly']['phases_ms']['solve_ms']:.2f}  launches/it {d['launches_per_iteration']:.1f}  each {d['each_timed_step']['ms']}"
    for k in ("config_c2", "config_c4"):
        if k in d:
            s += f"\n             {k}: {d[k]['ms_per_step']:.2f} ms  its {d[k]['iterations']}  solve {d[k]['phases_ms']['solve_ms']:.2f}"
    print(s)
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "fused_cycle or amg_preconditioner or amg_" 2>&1 | tail -12
PFV_AMG_FUSE_CYCLE=1 run fused
PFV_AMG_FUSE_CYCLE=0 run unfused
PFV_AMG_FUSE_CYCLE=1 run fused_b --no-extra-configs
PFV_AMG_FUSE_CYCLE=0 run unfused_b --no-extra-configs
