#!/bin/bash
# Round 5, call E: the whole GPU suite under PFV_NODE_GJ=5 (unpivoted + verified elimination), node kernel lab.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5e
mkdir -p $O
python tools/node_lab.py 69 2>&1 | tail -8 | tee $O/node_lab.log
PFV_NODE_GJ=5 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_gj5.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu_gj5.log; tail -6 $O/pytest_gpu_gj5.log
