#!/bin/bash
# Round 4, call D: AMG tests of the GPU suite on the current tree + A/B bench lines (name:"ENV=..").
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/${1:-r4d}; shift
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_block_solver.py -m gpu -q --timeout 600 -k "amg or block or fracture or solver" > $O/pytest_amg.log 2>&1
echo "pytest exit $?" >> $O/pytest_amg.log; tail -5 $O/pytest_amg.log
bash tools/gpu_r4_c.sh ${O#gpurun_out/} "$@"
