#!/bin/bash
# Round 6 lab: the 8-lane threshold of the f32 cycle products beyond 40 entries per row (does MPSA's hierarchy gain?)
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6lab7
mkdir -p $O
run() {
  local name=$1; shift
  env "$@" timeout 900 python bench.py --no-cpu-baseline --no-whole-grid-check --no-cold --steps 4 --warmup 2 \
    > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    s = f"{sys.argv[2]:8s} step {d['ms_per_step']:.2f} ms  solve {d['assembly']['phases_ms']['solve_ms']:.2f}"
    for k in ("config_c2", "config_c4"):
        if k in d:
            s += f" | {k}: {d[k]['ms_per_step']:.2f} ms its {d[k]['iterations']} solve {d[k]['phases_ms']['solve_ms']:.2f}"
    print(s)
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in a b; do
  for t in 40 60 100 160; do run l8_${t}_$rep PFV_SPMV_L8_MAX_F32=$t; done
done
