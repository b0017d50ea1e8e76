#!/bin/bash
# The N > 1 code path on one MI355X, coupled hierarchy (pfv_amg_setup_sharded) against the block hierarchy:
#   * one rank through the native RCCL hooks (communicator of one rank: ncclAllGather / ncclAllReduce run, no peers);
#   * two ranks sharing the GPU over gloo (torch.distributed transport of the packed buffers) -- iteration counts and
#     exchange counts; the times of two processes sharing one device are not scaling numbers.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r3s}
mkdir -p $O
for pc in amg amg_block; do
  timeout 300 python bench.py --force-sharded --precond $pc --no-cpu-baseline --no-extra-configs > $O/sharded_rccl_1rank_$pc.json 2> $O/sharded_rccl_1rank_$pc.err
  echo "1-rank native RCCL $pc exit $?"; tail -c 200 $O/sharded_rccl_1rank_$pc.err
done
for ns in 40 69; do
  for pc in amg amg_block; do
    PFV_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port 29517 bench.py --gpus 2 --n-side $ns --precond $pc --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs \
      > $O/gloo_2rank_n${ns}_$pc.json 2> $O/gloo_2rank_n${ns}_$pc.err
    echo "2-rank gloo n_side $ns $pc exit $?"; tail -c 300 $O/gloo_2rank_n${ns}_$pc.err | tail -2
  done
done
python bench.py --n-side 40 --no-cpu-baseline --no-extra-configs --steps 5 --warmup 2 > $O/single_n40.json 2> $O/single_n40.err
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d["config"]
        print(os.path.basename(f), "ms/step", round(d["ms_per_step"], 2), "its", c["iterations"], "cells", c.get("global_cells"),
              "levels", (c.get("amg") or {}).get("levels"), "coarsest", (c.get("amg") or {}).get("coarsest_rows"),
              "setup_ms", round((c.get("amg") or {}).get("setup_ms") or 0, 2), "transport", c.get("transport"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
