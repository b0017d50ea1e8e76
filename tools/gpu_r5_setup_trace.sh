#!/bin/bash
# kernel timeline of the AMG setup of a step on moving values (tools/run_step.py with PFV_RUN_STEP_MOVING=1)
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r5setup
mkdir -p $O
cd /tmp
export PFV_RUN_STEP_MOVING=1
python $R/tools/run_step.py > $O/plain.log 2>&1
rm -rf /tmp/r5trace
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/r5trace -o t --output-format csv -- python $R/tools/run_step.py > $O/traced.log 2>&1
python $R/tools/step_timeline.py /tmp/r5trace 0 > $O/timeline_full.txt 2>&1
python - "$O" <<'PY'
import sys
o = sys.argv[1]
L = open(o + "/timeline_full.txt").read().splitlines()
i0 = max(i for i, l in enumerate(L) if "assemble_system" in l)
i1 = min(i for i, l in enumerate(L) if i > i0 and "k_spmv_win_pre<16, 5, double" in l)
open(o + "/timeline_setup.txt", "w").write("\n".join(L[i0:i1 + 1]) + "\n")
print(len(L), "lines; setup part", i1 - i0)
PY
tail -2 $O/plain.log; tail -1 $O/timeline_full.txt
