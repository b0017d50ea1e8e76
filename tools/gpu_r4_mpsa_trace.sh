#!/bin/bash
# Kernel trace of the MPSA step at configs[3] size (tools/bench_mpsa.py 44) -> gpurun_out/mpsa_kernel_stats.txt
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out
mkdir -p $O; cd /tmp; rm -rf $O/mtrace
timeout 600 rocprofv3 --kernel-trace --stats -d $O/mtrace -o t --output-format csv -- python $R/tools/bench_mpsa.py 44 > $O/mpsa_traced.log 2>&1
python - <<'PY'
import csv, os, re
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
rows = list(csv.DictReader(open(os.path.join(root, "gpurun_out/mtrace/t_kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = [f"rocprofv3 --kernel-trace --stats -- python tools/bench_mpsa.py 44   (2 discretizations + 1 solve); total kernel time {tot/1e6:.1f} ms",
       f"{'total_ms':>10} {'pct':>6} {'calls':>7} {'avg_us':>10}  kernel"]
for r in rows[:60]:
    name = re.sub(r"\(pfv::pfv_ctx_impl&[^)]*\)", "()", r["Name"])
    out.append(f'{float(r["TotalDurationNs"])/1e6:10.2f} {100*float(r["TotalDurationNs"])/tot:6.2f} {int(r["Calls"]):7d} {float(r["AverageNs"])/1e3:10.1f}  {name[:150]}')
open(os.path.join(root, "gpurun_out/mpsa_kernel_stats.txt"), "w").write("\n".join(out) + "\n")
PY
rm -rf $O/mtrace
tail -8 $O/mpsa_traced.log
