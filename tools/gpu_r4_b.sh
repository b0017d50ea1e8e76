#!/bin/bash
# Round 4, call B: the 8 x 8 lane-grid Gauss-Jordan (PFV_NODE_GJ=3) against the lane = row form: parity tests at full
# size under the switch, node-kernel time per variant.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r4b
mkdir -p $O
PFV_NODE_GJ=3 timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "timed_bench_grid or config_c2 or sliver or larger_than_lds or golden_case" > $O/pytest_gj3.log 2>&1
echo "pytest exit $?" >> $O/pytest_gj3.log; tail -5 $O/pytest_gj3.log
PFV_LAB_NODE="2,3,2,3" timeout 600 python tools/face_lab.py 69 only base > $O/node_lab.log 2> $O/node_lab.err
cat $O/node_lab.log; tail -3 $O/node_lab.err
