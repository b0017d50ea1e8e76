#!/bin/bash
# Round 5, late: the symbolic phase is the critical path beside the 6.8 ms node kernel -- re-measure the deferred cell rows
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5o
mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra-configs --no-whole-grid-check --no-cold > $O/$tag.json 2> $O/$tag.err
  python - "$O/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"{sys.argv[2]:20s} ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} asm {d['assembly']['ms']:.2f} {ph}")
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
}
run default PFV_X=0
run defer_cells PFV_SYMB_DEFER_CELLS=1
run node_lowprio PFV_NODE_LOWPRIO=1
run defer_lowprio PFV_SYMB_DEFER_CELLS=1 PFV_NODE_LOWPRIO=1
run default2 PFV_X=0
