#!/bin/bash
# One gpurun call for the assembly kernels: bash tools/gpu_lab.sh [--tests] label:"ENV=1 ..." ...
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
mkdir -p $O
: > $O/lab.log
if [ "$1" == "--tests" ]; then
  shift
  timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $O/pytest_gpu.log
  tail -15 $O/pytest_gpu.log
fi
for v in "$@"; do
  name="${v%%:*}"
  envs="${v#*:}"
  env $envs timeout 300 python tools/asm_lab.py "$name" >> $O/lab.log 2> $O/lab_$name.err || { echo "$name FAILED"; tail -5 $O/lab_$name.err; }
done
cat $O/lab.log
