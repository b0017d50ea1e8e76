#!/bin/bash
# Round 5, call F: MPSA elimination without the pivot search (PFV_MPSA_GJ_NP=1): parity under the switch, configs[3] A/B;
# the whole-grid pattern datum.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5f
mkdir -p $O
python - > $O/c4_ab.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench, porepy_amd as pa
for np_ in ("0", "1"):
    os.environ["PFV_MPSA_GJ_NP"] = np_
    r = bench.bench_config_c4(pa, 0, 1e-13, "amg", steps=2)
    print("PFV_MPSA_GJ_NP", np_, "ms/step %.1f its %d err %.2e" % (r["ms_per_step"], r["iterations"], r["max_abs_error_vs_exact_uniaxial_field"]), {k: round(v, 1) for k, v in r["phases_ms"].items()}, flush=True)
PY
cat $O/c4_ab.log
PFV_MPSA_GJ_NP=1 timeout 900 python -m pytest tests/test_gpu_mpsa.py -m gpu -q -x --timeout 800 > $O/pytest_mpsa_np.log 2>&1
echo "pytest exit $?" >> $O/pytest_mpsa_np.log; tail -5 $O/pytest_mpsa_np.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 500 -k "whole_headline or sliver" > $O/pytest_whole.log 2>&1
echo "pytest exit $?" >> $O/pytest_whole.log; tail -5 $O/pytest_whole.log
