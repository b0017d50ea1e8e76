#!/bin/bash
# Round 6 lab: 8 lanes per row for the f32 cycle products up to 40 entries per row (PFV_SPMV_L8_MAX_F32=40, adopted) against 20
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6lab6
mkdir -p $O
run() {
  local name=$1; shift
  env "$@" timeout 900 python bench.py --no-cpu-baseline --no-whole-grid-check --no-cold --steps 6 --warmup 2 \
    > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    s = f"{sys.argv[2]:8s} step {d['ms_per_step']:.2f} ms  its {d['each_timed_step']['iterations']}  solve {d['assembly']['phases_ms']['solve_ms']:.2f}  each {d['each_timed_step']['ms']}"
    for k in ("config_c2", "config_c4"):
        if k in d:
            s += f"\n         {k}: {d[k]['ms_per_step']:.2f} ms  its {d[k]['iterations']}  solve {d[k]['phases_ms']['solve_ms']:.2f}"
    print(s)
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run l8_40 PFV_SPMV_L8_MAX_F32=40
run l8_20 PFV_SPMV_L8_MAX_F32=20
run l8_40b PFV_SPMV_L8_MAX_F32=40
run l8_20b PFV_SPMV_L8_MAX_F32=20
run l8_60 PFV_SPMV_L8_MAX_F32=60
