#!/bin/bash
# Kernel trace of the default bench step -> gpurun_out/bench_kernel_stats.txt (+ the traced bench line)
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out
mkdir -p $O
cd /tmp
rm -rf $O/btrace
timeout 400 rocprofv3 --kernel-trace --stats -d $O/btrace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra-configs --steps 3 --warmup 1 > $O/bench_traced.json 2> $O/btrace.log
python - <<'PY'
import csv, os, re
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
p = os.path.join(root, "gpurun_out/btrace/t_kernel_stats.csv")
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = [f"rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extra-configs --steps 3 --warmup 1",
       f"total kernel time {tot/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} dispatches (pre-warm + warmup + 3 timed steps + kernel timing loops)",
       f"{'total_ms':>10} {'pct':>6} {'calls':>7} {'avg_us':>10} {'min_us':>9} {'max_us':>10}  kernel"]
small = [0, 0.0]
for r in rows:
    name = r["Name"]
    name = re.sub(r"\(pfv::pfv_ctx_impl&[^)]*\)", "()", name)
    avg = float(r["AverageNs"]) / 1e3
    if avg < 15.0:
        small[0] += int(r["Calls"]); small[1] += float(r["TotalDurationNs"]) / 1e6
    out.append(f'{float(r["TotalDurationNs"])/1e6:10.2f} {100*float(r["TotalDurationNs"])/tot:6.2f} {int(r["Calls"]):7d} {avg:10.1f} {float(r["MinNs"])/1e3:9.1f} {float(r["MaxNs"])/1e3:10.1f}  {name[:170]}')
out.insert(2, f"kernels with an average duration below 15 us: {small[0]} dispatches, {small[1]:.2f} ms in total")
open(os.path.join(root, "gpurun_out/bench_kernel_stats.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:45]))
PY
tail -c 400 $O/bench_traced.json | head -c 10; echo
