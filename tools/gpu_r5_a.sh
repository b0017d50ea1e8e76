#!/bin/bash
# Round 5, call A: the new GPU tests (whole headline grid vs the recorded reference run, MPSA reference-on-patches, the
# batch path with vector sources), SQ / traffic counters of one full step INCLUDING the interaction-region kernel
# (tools/pmc_summary.py now keeps its rows), and bench lines: new field per step (default), one field repeated, cold.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5a
mkdir -p $O
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $O/timeline.log; }
timeout 900 python -m pytest tests -m gpu -q -x --timeout 800 -k "whole_headline or config_c3_all_four or batch_matches_single or device_build_loaded" > $O/pytest_subset.log 2>&1
echo "pytest exit $?" >> $O/pytest_subset.log; tail -5 $O/pytest_subset.log; stamp tests
timeout 420 python bench.py --no-cpu-baseline --no-extra-configs --no-whole-grid-check --steps 10 > $O/bench_moving.json 2> $O/bench_moving.err; stamp bench_moving
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --no-whole-grid-check --steps 10 --fixed-k > $O/bench_fixed.json 2> $O/bench_fixed.err; stamp bench_fixed
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
for tag in ("moving", "fixed"):
    try:
        d = json.loads([l for l in open(f"{o}/bench_{tag}.json") if l.startswith("{")][-1])
        ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
        print(f"{tag}: ms/step {d['ms_per_step']:.2f} cold {d['ms_per_step_cold']:.2f} its {d['config']['iterations']} cold its {d['cold_step']['iterations']} "
              f"asm {d['assembly']['ms']:.2f} {ph} amg_setup {d['config']['amg']['setup_ms']:.2f} launches/it {d['launches_per_iteration']:.1f} reuse {d['config']['pattern_reuse']['amg_aggregate_maps_kept']}")
    except Exception as e:
        print(tag, "bench FAILED", e, open(f"{o}/bench_{tag}.err").read()[-1500:])
PY
bash tools/gpu_pmc.sh > $O/pmc_stdout.log 2>&1
cp gpurun_out/pmc_summary.txt $O/pmc_summary.txt; cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json
grep -A12 "run[3-5]" $O/pmc_summary.txt | grep -E "run|node|symb|face" | cut -c1-1200
stamp pmc
cat $O/timeline.log
