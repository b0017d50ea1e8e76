#!/bin/bash
# PMC passes over one full step (tools/run_step.py): counters in their own runs, kernel trace apart.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out
rm -rf $O/pmc; mkdir -p $O/pmc
cd /tmp
python $R/tools/run_step.py > $O/pmc/plain.log 2>&1   # builds the grid cache
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "$@"; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $O/pmc/run$i -o c --output-format csv -- python $R/tools/run_step.py > $O/pmc/run$i.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
python $R/tools/pmc_to_json.py $O/pmc $O/pmc_traffic.json > $O/pmc_to_json.log 2>&1
rm -rf $O/pmc   # per-dispatch CSVs of five passes: too large to carry back (gpurun_out is capped at 64 MiB)
cut -c1-400 $O/pmc_summary.txt | grep -v "win <16, [34]" | head -60
cat $O/pmc_traffic.json
