#!/bin/bash
# Round 6 lab: XCD-aware order of the row blocks of the windowed SpMV (PFV_SPMV_XCD) against the plain blockIdx order
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6lab4
mkdir -p $O
run() {
  local name=$1; shift
  env "$@" timeout 900 python bench.py --no-cpu-baseline --no-whole-grid-check --no-cold --steps 6 --warmup 2 \
    > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    rk = {r.get("name"): r for r in d["roofline_kernels"]}
    f = rk["amg_f32_smoothing_product"]
    s = (f"{sys.argv[2]:10s} step {d['ms_per_step']:.2f} ms  its {d['each_timed_step']['iterations']}  solve {d['assembly']['phases_ms']['solve_ms']:.2f}  "
         f"f64 product {d['roofline']['ms_per_launch'] * 1e3:.1f} us (frac {d['roofline']['frac']:.3f})  f32 product {f['ms_per_launch'] * 1e3:.1f} us (frac {f['frac']:.3f})")
    for k in ("config_c2", "config_c4"):
        if k in d:
            s += f"\n           {k}: {d[k]['ms_per_step']:.2f} ms  its {d[k]['iterations']}  solve {d[k]['phases_ms']['solve_ms']:.2f}"
    print(s)
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run xcd1 PFV_SPMV_XCD=1
run xcd0 PFV_SPMV_XCD=0
run xcd1_b PFV_SPMV_XCD=1
run xcd0_b PFV_SPMV_XCD=0
