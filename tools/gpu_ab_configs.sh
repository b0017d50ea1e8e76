#!/bin/bash
# A/B of the three configurations (headline, configs[1], configs[3]): bash tools/gpu_ab_configs.sh name:"ENV=.." ...
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
mkdir -p $O
for v in "$@"; do
  n="${v%%:*}"; e="${v#*:}"
  env $e timeout 400 python bench.py --no-cpu-baseline > $O/abc_$n.json 2> $O/abc_$n.err
  python - "$n" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads([l for l in open(f"gpurun_out/abc_{n}.json") if l.startswith("{")][-1])
    c2, c4 = d["config_c2"], d["config_c4"]
    print(f"{n:10s} C3 {d['ms_per_step']:.2f} ms its {d['config']['iterations']} solve {d['assembly']['phases_ms']['solve_ms']:.1f} | "
          f"C2 {c2['ms_per_step']:.2f} ms its {c2['iterations']} solve {c2['phases_ms']['solve_ms']:.2f} | "
          f"C4 {c4['ms_per_step']:.1f} ms its {c4['iterations']} solve {c4['phases_ms']['solve_ms']:.1f}")
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/abc_{n}.err").read()[-500:])
PY
done
