#!/bin/bash
# Round 6 lab: the f32 products of the AMG cycle's finest level (k_spmv_win<8, 3, float>: 0.105 ms, 0.42 of the HBM roofline
# by CSR bytes) under the switches of spmv_win.inc -- preloading form, 8 vs 16 lanes per row, chunks per lane.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6lab
mkdir -p $O
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --no-whole-grid-check --no-cold --steps 6 --warmup 2 \
    > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernel_ms_per_step"]
    rk = {r.get("name"): r for r in d["roofline_kernels"]}
    f = rk["amg_f32_smoothing_product"]
    print(f"{sys.argv[2]:28s} step {d['ms_per_step']:.2f} ms  its {d['config']['iterations']}  solve {d['assembly']['phases_ms']['solve_ms']:.2f}  "
          f"f32 product {f['ms_per_launch'] * 1e3:.1f} us x {f['launches_per_step']}  f64 product {d['roofline']['ms_per_launch'] * 1e3:.1f} us  "
          f"each {d['each_timed_step']['ms']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run base
run pre32 PFV_SPMV_PRELOAD_F32=1
run l16u2 PFV_SPMV_L8_MAX=16
run l16u2_pre32 PFV_SPMV_L8_MAX=16 PFV_SPMV_PRELOAD_F32=1
run u2 PFV_SPMV_U_F32=2
run u2_pre32 PFV_SPMV_U_F32=2 PFV_SPMV_PRELOAD_F32=1
run u4 PFV_SPMV_U_F32=4
run base_again
