#!/bin/bash
# Round 6, checkpoint D: rows of the 8 M-cell one-handle matrices against the split path; cost of the double-double MPSA body.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6d
mkdir -p $O
timeout 1500 python tools/big_handle_check.py 110 24 > $O/big_handle_110.json 2> $O/big_handle_110.err
tail -c 2500 $O/big_handle_110.json; tail -c 400 $O/big_handle_110.err
timeout 900 python tools/mpsa_dd_cost.py 24 > $O/mpsa_dd_cost.txt 2> $O/mpsa_dd_cost.err
cat $O/mpsa_dd_cost.txt; tail -c 400 $O/mpsa_dd_cost.err
