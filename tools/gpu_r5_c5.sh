#!/bin/bash
# Round 5: configs[4] stand-in at 32^3 (36 657 cells, 98 229 unknowns): device-backed model and the untouched reference.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r5c5
C5_N_SIDE=32 C5_MAX_EXTENT=14 PFV_DROPIN_LIBRARY=product timeout 2400 python tools/c5_bench.py --reference > gpurun_out/r5c5/c5_32.log 2>&1
tail -2 gpurun_out/r5c5/c5_32.log | cut -c1-3000
