#!/bin/bash
# Kernel trace of the assembly phases (tools/run_asm.py) -> gpurun_out/trace_stats.txt (top kernels)
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out
mkdir -p $O
cd /tmp
rm -rf $O/trace
env "$@" timeout 240 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/tools/run_asm.py > $O/trace.log 2>&1
python - <<'PY'
import csv, os, re
p = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/trace/t_kernel_stats.csv")
rows = list(csv.DictReader(open(p)))
out = []
for r in rows[:32]:
    name = r["Name"]
    name = re.sub(r"pfv::k_(wave_for|parallel_for|wave_for_xcd|block_for)", r"\1", name)
    name = re.sub(r"\(pfv::pfv_ctx_impl&[^)]*\)", "()", name)
    out.append(f'{float(r["TotalDurationNs"])/1e6:9.3f} ms  n={int(r["Calls"]):4d}  avg={float(r["AverageNs"])/1e3:9.1f} us  {name[:150]}')
open(os.path.join(os.path.dirname(os.path.dirname(p)), "trace_stats.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -2 $O/trace.log
