#!/bin/bash
# A/B bench runs on one MI355X box: bash tools/gpu_ab.sh [--tests] name1:"ENV=1 ENV2=x" name2:"" ...
# (each variant: python bench.py --no-cpu-baseline --no-extra-configs with that environment)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
mkdir -p $O
if [ "$1" == "--tests" ]; then
  shift
  timeout 600 python -m pytest tests -m gpu -q --timeout 180 > $O/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $O/pytest_gpu.log
  tail -4 $O/pytest_gpu.log
fi
for v in "$@"; do
  name="${v%%:*}"
  envs="${v#*:}"
  env $envs timeout 200 python bench.py --no-cpu-baseline --no-extra-configs > $O/ab_$name.json 2> $O/ab_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"{n:14s} ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} res {d['config']['true_rel_residual']:.2e} "
          f"amg_setup {d['config']['amg']['setup_ms']:.2f} {ph}")
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/ab_{n}.err").read()[-800:])
PY
done
