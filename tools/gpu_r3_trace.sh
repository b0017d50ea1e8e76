#!/bin/bash
# kernel traces of the bench step under environment switches: bash tools/gpu_r3_trace.sh OUTDIR name:"ENV=.." ...
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp
for v in "$@"; do
  n="${v%%:*}"; e="${v#*:}"
  rm -rf $O/tr_$n
  env $e timeout 400 rocprofv3 --kernel-trace --stats -d $O/tr_$n -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra-configs --steps 3 --warmup 1 > $O/traced_$n.json 2> $O/trace_$n.log
  python - "$O" "$n" <<'PY'
import csv, os, re, sys, glob
o, n = sys.argv[1], sys.argv[2]
p = glob.glob(os.path.join(o, f"tr_{n}", "**", "t_kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = [f"rocprofv3 --kernel-trace --stats -- [{n}] python bench.py --no-cpu-baseline --no-extra-configs --steps 3 --warmup 1",
       f"total kernel time {tot/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} dispatches (pre-warm + warmup + 3 timed steps + kernel timing loops)",
       f"{'total_ms':>10} {'pct':>6} {'calls':>7} {'avg_us':>10} {'min_us':>9} {'max_us':>10}  kernel"]
small = [0, 0.0]
for r in rows:
    name = re.sub(r"\(pfv::pfv_ctx_impl&[^)]*\)", "()", r["Name"])
    avg = float(r["AverageNs"]) / 1e3
    if avg < 15.0:
        small[0] += int(r["Calls"]); small[1] += float(r["TotalDurationNs"]) / 1e6
    out.append(f'{float(r["TotalDurationNs"])/1e6:10.2f} {100*float(r["TotalDurationNs"])/tot:6.2f} {int(r["Calls"]):7d} {avg:10.1f} {float(r["MinNs"])/1e3:9.1f} {float(r["MaxNs"])/1e3:10.1f}  {name[:170]}')
out.insert(2, f"kernels with an average duration below 15 us: {small[0]} dispatches, {small[1]:.2f} ms in total")
open(os.path.join(o, f"kernel_stats_{n}.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
PY
  find $O/tr_$n -type f -size +20M -delete
done
