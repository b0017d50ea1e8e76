#!/bin/bash
# Round 6: kernel timeline of one BiCGStab iteration of the headline solve with the fused cycle (rocprofv3 --kernel-trace of
# tools/run_step.py on moving values; the counterpart of profiles/r05_iteration_timeline.txt)
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r6iter
mkdir -p $O
cd /tmp
export PFV_RUN_STEP_MOVING=1
rm -rf /tmp/r6trace
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/r6trace -o t --output-format csv -- python $R/tools/run_step.py > $O/traced.log 2>&1
python $R/tools/step_timeline.py /tmp/r6trace 0 > $O/timeline_full.txt 2>&1
python - "$O" <<'PY'
import sys
o = sys.argv[1]
L = open(o + "/timeline_full.txt").read().splitlines()
# the iterations of the LAST solve: every iteration starts with the product with the dots of mode 2
starts = [i for i, l in enumerate(L) if "k_spmv_win_pre<16, 5, double, 2>" in l]
print(len(L), "lines;", len(starts), "iteration starts")
k = len(starts) - 18 if len(starts) > 20 else len(starts) // 2  # iteration 6 of the last solve (24 iterations)
open(o + "/timeline_iteration.txt", "w").write("\n".join(L[starts[k]:starts[k + 1]]) + "\n")
print("dispatches in the iteration:", starts[k + 1] - starts[k])
PY
tail -2 $O/traced.log; head -3 $O/timeline_iteration.txt
