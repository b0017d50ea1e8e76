#!/bin/bash
# Round 4, call C: full GPU suite + bench line of the current tree (+ optional A/B switches as name:"ENV=..").
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/${1:-r4c}; shift
mkdir -p $O
if [ "$1" == "--alltests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=8 > $O/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log; shift
fi
STEPS=5
for v in "$@"; do
  n="${v%%:*}"; e="${v#*:}"
  env $e timeout 400 python bench.py --no-cpu-baseline --no-extra-configs --steps $STEPS > $O/ab_$n.json 2> $O/ab_$n.err
  python - "$O" "$n" <<'PY'
import json, sys
o, n = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"{o}/ab_{n}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"{n:14s} ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} res {d['config']['true_rel_residual']:.2e} "
          f"amg_setup {d['config']['amg']['setup_ms']:.2f} lev {d['config']['amg']['levels']} asm {d['assembly']['ms']:.2f} {ph}")
except Exception as e:
    print(n, "FAILED", e, open(f"{o}/ab_{n}.err").read()[-900:])
PY
done
