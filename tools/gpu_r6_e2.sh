#!/bin/bash
# Round 6, last library: the N > 1 path of bench.py once more on the one-GPU box (reduced tools/gpu_r6_e.sh: one rank through the
# sharded path over RCCL; two ranks sharing GPU 0 over gloo, weak and strong) -- the coupled hierarchy's replicated part runs the
# fused cycle too
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6e2
mkdir -p $O
show() {
python - "$1" <<'PY'
import json, sys
p = sys.argv[1]
try:
    d = json.loads([l for l in open(p) if l.startswith("{")][-1])
    print(p.split("/")[-1], "n_gpus", d["n_gpus"], "scaling", d["scaling"], "ms/step", round(d["ms_per_step"], 2), "value", f"{d['value']:.3e}", "its", d["config"]["iterations"],
          "cells", d["config"]["global_cells"], "backend", d["config"].get("process_group_backend"), "residual", d["config"].get("true_rel_residual"))
    for r in (d.get("per_rank") or []):
        if "error" in r:
            print("   per_rank error", r["error"]); continue
        print("   rank", r["rank"], "owned", r["cells_owned"], "phases", {k[:-3]: round(v, 2) for k, v in r["phases_ms"].items()}, "kept", r["kept"], "seen", r["ranks_seen_by_rccl"], r["backend"])
except Exception as e:
    print(p, "FAILED", e, open(p.replace(".json", ".err")).read()[-1500:])
PY
}
timeout 900 python bench.py --force-sharded --steps 4 --warmup 2 --no-cold --no-cpu-baseline --no-whole-grid-check --no-extra-configs > $O/sharded_one_rank_rccl.json 2> $O/sharded_one_rank_rccl.err
show $O/sharded_one_rank_rccl.json
for cfg in "2:weak:48" "2:strong:69"; do
  n=${cfg%%:*}; rest=${cfg#*:}; sc=${rest%%:*}; ns=${rest#*:}
  PFV_BENCH_SHARE_GPU=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n + ${#sc})) \
      bench.py --gpus $n --scaling $sc --n-side $ns --steps 3 --warmup 1 --no-cold --no-cpu-baseline --no-whole-grid-check --no-extra-configs > $O/share_${n}_${sc}.json 2> $O/share_${n}_${sc}.err
  show $O/share_${n}_${sc}.json
done
