#!/bin/bash
# kernel timeline of one step with the interaction-region kernel AFTER the symbolic phase (PFV_OVERLAP_NODE=0):
# what every symbolic kernel takes when it has the machine to itself
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r5t
mkdir -p $O
cd /tmp
export PFV_OVERLAP_NODE=${PFV_OVERLAP_NODE:-0}
python $R/tools/run_step.py > $O/plain.log 2>&1
rm -rf /tmp/r5trace
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/r5trace -o t --output-format csv -- python $R/tools/run_step.py > $O/traced.log 2>&1
python $R/tools/step_timeline.py /tmp/r5trace 30 > $O/timeline_serial.txt 2>&1
grep -n "k_face_pipe" $O/timeline_serial.txt | head -2
head -110 $O/timeline_serial.txt | cut -c1-200
