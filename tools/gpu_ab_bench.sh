#!/bin/bash
# A/B bench lines on one box: bash tools/gpu_ab_bench.sh [--tests "<pytest -k expr>"] name:"ENV=.." ...
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
mkdir -p $O
if [ "$1" == "--tests" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "$2" > $O/pytest_gpu_subset.log 2>&1
  echo "pytest exit $?" >> $O/pytest_gpu_subset.log
  tail -4 $O/pytest_gpu_subset.log
  shift; shift
fi
for v in "$@"; do
  n="${v%%:*}"; e="${v#*:}"
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $O/ab_$n.json 2> $O/ab_$n.err
  python - "$n" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads([l for l in open(f"gpurun_out/ab_{n}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"{n:12s} ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} res {d['config']['true_rel_residual']:.2e} "
          f"amg_setup {d['config']['amg']['setup_ms']:.2f} {ph}")
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/ab_{n}.err").read()[-600:])
PY
done
