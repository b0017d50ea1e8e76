#!/bin/bash
# kernel timeline of the last step of the sharded path on one rank (bench.py --force-sharded, K moving): setup part
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r5shard
mkdir -p $O
cd $R
rm -rf /tmp/r5s
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r5s -o t --output-format csv -- python bench.py --force-sharded --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --no-whole-grid-check --no-cold > $O/traced.json 2> $O/traced.err
python tools/step_timeline.py /tmp/r5s 0 > $O/timeline_full.txt 2>&1
python - "$O" <<'PY'
import sys
o = sys.argv[1]
L = open(o + "/timeline_full.txt").read().splitlines()
i0 = max(i for i, l in enumerate(L) if "assemble_system" in l)
open(o + "/timeline_after_assemble.txt", "w").write("\n".join(L[i0:i0 + 700]) + "\n")
print(len(L), "lines")
PY
tail -1 $O/timeline_full.txt; tail -c 300 $O/traced.err
