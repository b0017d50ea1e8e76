#!/bin/bash
# kernel timeline of one step (tools/run_step.py) -> gpurun_out/r4t/timeline.txt (+ stats of the run)
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r4t
mkdir -p $O
cd /tmp
python $R/tools/run_step.py > $O/plain.log 2>&1
rm -rf /tmp/r4trace
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/r4trace -o t --output-format csv -- python $R/tools/run_step.py > $O/traced.log 2>&1
python $R/tools/step_timeline.py /tmp/r4trace ${1:-0} > $O/timeline.txt 2>&1
python $R/tools/trace_top.py $(find /tmp/r4trace -name "*kernel_stats.csv" | head -1) 60 > $O/kernel_stats_top.txt 2>&1
head -c 60000 $O/timeline.txt | head -400
tail -3 $O/timeline.txt
