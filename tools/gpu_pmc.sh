#!/bin/bash
# PMC passes over the assembly phases (tools/run_asm.py): counters in their own runs, kernel trace apart.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out
mkdir -p $O/pmc
cd /tmp
python $R/tools/run_asm.py > $O/pmc/plain.log 2>&1   # builds the grid cache
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "$@"; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set -d $O/pmc/run$i -o c --output-format csv -- python $R/tools/run_asm.py > $O/pmc/run$i.log 2>&1
done
(timeout 240 rocprofv3 --kernel-trace --stats -d $O/pmc/trace -o t --output-format csv -- python $R/tools/run_asm.py > $O/pmc/trace.log 2>&1)
python $R/tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
ls $O/pmc/trace/* | head
