#!/bin/bash
# Round 6 lab (side build with PFV_SPMV_PRELOAD_F32 / PFV_SPMV_L8_MAX / PFV_SPMV_U_F32): the f32 products of the COARSE levels
# (250 k and 31 k rows: 2 rounds of workgroups, where a block's chain of dependent loads is what is waited for) per kernel,
# from rocprofv3 --kernel-trace --stats of tools/run_step.py
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r6lab5
mkdir -p $O
cd /tmp
export PFV_RUN_STEP_MOVING=1
run() {
  local name=$1; shift
  rm -rf /tmp/r6c
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r6c -o t --output-format csv -- python $R/tools/run_step.py > $O/$name.log 2>&1
  python - "$name" <<'PY'
import csv, glob, sys, re
f = glob.glob("/tmp/r6c/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = 0.0
out = []
for r in rows:
    n = r["Name"]
    if "k_spmv_win" in n and "float" in n:
        m = re.search(r"k_spmv_win(_pre)?<(\d+), (\d+), float, (\d+)>", n)
        key = ("pre " if m.group(1) else "    ") + f"L{m.group(2)} U{m.group(3)} mode{m.group(4)}"
        out.append((key, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3 / int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6))
        tot += float(r["TotalDurationNs"]) / 1e6
print(f"== {sys.argv[1]}: all f32 windowed products {tot:.2f} ms")
for k in sorted(out):
    print(f"   {k[0]:22s} calls {k[1]:5d}  avg {k[2]:7.1f} us  total {k[3]:7.2f} ms")
PY
  grep -i "iterations\|ms" $O/$name.log | tail -2 | cut -c1-200
}
run base PFV_LAB=0
run pre32 PFV_SPMV_PRELOAD_F32=1
run l8_30 PFV_SPMV_L8_MAX=40
run l8_30_pre PFV_SPMV_L8_MAX=40 PFV_SPMV_PRELOAD_F32=1
run u1 PFV_SPMV_U_F32=1
run u4 PFV_SPMV_U_F32=4
