#!/bin/bash
# Round 3 A/B call: GPU suite of the current tree, then bench lines under environment switches.
#   bash tools/gpu_r3_ab.sh OUTDIR [--tests "<pytest -k expr>"|--alltests] [--steps K] name:"ENV=.. ENV2=.." ...
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$1; shift
mkdir -p $O
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $O/timeline.log; }
if [ "$1" == "--tests" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "$2" > $O/pytest_gpu_subset.log 2>&1
  echo "pytest exit $?" >> $O/pytest_gpu_subset.log; tail -4 $O/pytest_gpu_subset.log; shift; shift; stamp tests
elif [ "$1" == "--alltests" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --timeout 400 --durations=10 > $O/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log; shift; stamp alltests
fi
STEPS=5
if [ "$1" == "--steps" ]; then STEPS=$2; shift; shift; fi
for v in "$@"; do
  n="${v%%:*}"; e="${v#*:}"
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps $STEPS > $O/ab_$n.json 2> $O/ab_$n.err
  stamp "$n"
  python - "$O" "$n" <<'PY'
import json, sys
o, n = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"{o}/ab_{n}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"{n:14s} ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} res {d['config']['true_rel_residual']:.2e} "
          f"amg_setup {d['config']['amg']['setup_ms']:.2f} lev {d['config']['amg']['levels']} asm {d['assembly']['ms']:.2f} {ph}")
except Exception as e:
    print(n, "FAILED", e, open(f"{o}/ab_{n}.err").read()[-600:])
PY
done
cat $O/timeline.log
