#!/bin/bash
# Round 5, call G: MPSA unpivoted elimination -- acceptance tolerance sweep (how many regions go to the redo list).
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5g
mkdir -p $O
python - > $O/c4_tol.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench, porepy_amd as pa
for np_, tolx in (("0", "1"), ("1", "1"), ("1", "10"), ("1", "100"), ("1", "10000"), ("1", "100000000")):
    os.environ["PFV_MPSA_GJ_NP"] = np_
    os.environ["PFV_MPSA_NP_TOLX"] = tolx
    r = bench.bench_config_c4(pa, 0, 1e-13, "amg", steps=1)
    ctx = pa.Context(0)
    print("NP", np_, "TOLX", tolx, "ms/step %.1f its %d err %.2e node %.1f" % (r["ms_per_step"], r["iterations"], r["max_abs_error_vs_exact_uniaxial_field"], r["phases_ms"]["node_ms"]), "redo", r.get("node_redo"), flush=True)
PY
cat $O/c4_tol.log
