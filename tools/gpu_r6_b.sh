#!/bin/bash
# Round 6, checkpoint B: the node || face pipeline (tests, bench under PFV_PIPE / PFV_PIPE_CHUNKS), and where the
# whole-grid MPSA digests at 511 104 cells moved from 2e-15 to 1e-13 (elimination variants).
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6b
mkdir -p $O
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $O/timeline.log; }
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "pipeline or rebuilt_topology or golden_case or deterministic or mid_size or partial or amg_preconditioner or timed_bench_grid" > $O/pytest_subset.log 2>&1
echo "pytest exit $?" >> $O/pytest_subset.log; tail -6 $O/pytest_subset.log; stamp pytest
for v in "default:" "pipe0:PFV_PIPE=0" "chunks4:PFV_PIPE_CHUNKS=4" "chunks16:PFV_PIPE_CHUNKS=16" "chunks30:PFV_PIPE_CHUNKS=30"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout 600 python bench.py --steps 8 --warmup 3 --no-cold --no-cpu-baseline --no-whole-grid-check --no-extra-configs > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O" "$name" <<'PY'
import json, sys
o, name = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"{o}/bench_{name}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"{name}: ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} asm {d['assembly']['ms']:.2f} frac {d['assembly']['frac_of_hbm_peak']:.3f} {ph}")
except Exception as e:
    print(name, "FAILED", e, open(f"{o}/bench_{name}.err").read()[-800:])
PY
done
stamp bench
for v in "default:" "pivoted:PFV_MPSA_GJ_NP=0" "notlean:PFV_MPSA_LEAN=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout 600 python - <<'PY' 2>&1 | tail -3
import sys, os
sys.path.insert(0, ".")
import porepy_amd as pa
from tests import _parity as P
lib = pa._lib.product_library()
out = P.mpsa_whole_grid_check(lib, 44)
print(os.environ.get("PFV_MPSA_GJ_NP"), os.environ.get("PFV_MPSA_LEAN"), {k: [float(f"{x:.2e}") for x in out[k]] for k in P.MPSA_KEYS})
PY
done
stamp mpsa
cat $O/timeline.log
