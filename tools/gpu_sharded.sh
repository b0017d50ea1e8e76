#!/bin/bash
# The N > 1 code path on one MI355X: one rank through the native RCCL hooks, two ranks sharing the GPU over
# gloo (Python hooks), and the attempt to run RCCL itself with two ranks on the one device.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
mkdir -p $O
timeout 300 python bench.py --force-sharded --no-cpu-baseline --no-extra-configs > $O/bench_sharded_rccl_1rank.json 2> $O/bench_sharded_rccl_1rank.err
echo "1-rank native RCCL exit $?"; tail -c 300 $O/bench_sharded_rccl_1rank.err
PFV_SHARDED_TRANSPORT=torch timeout 300 python bench.py --force-sharded --no-cpu-baseline --no-extra-configs > $O/bench_sharded_torch_hooks_1rank.json 2> $O/bench_sharded_torch_hooks_1rank.err
echo "1-rank torch hooks exit $?"
PFV_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29517 bench.py --gpus 2 --n-side 40 --no-cpu-baseline --no-extra-configs > $O/bench_2rank_gloo_shared_gpu.json 2> $O/bench_2rank_gloo_shared_gpu.err
echo "2-rank gloo exit $?"
PFV_BENCH_SHARE_GPU=rccl NCCL_DEBUG=WARN timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29519 bench.py --gpus 2 --n-side 40 --no-cpu-baseline --no-extra-configs > $O/bench_2rank_rccl_shared_gpu.json 2> $O/bench_2rank_rccl_shared_gpu.err
echo "2-rank RCCL on one device exit $?"; grep -i -m5 "error\|duplicate\|invalid" $O/bench_2rank_rccl_shared_gpu.err | cut -c1-300
python - <<'PY'
import json
for n in ("bench_sharded_rccl_1rank", "bench_sharded_torch_hooks_1rank", "bench_2rank_gloo_shared_gpu", "bench_2rank_rccl_shared_gpu"):
    try:
        d = json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
        print(n, "ms/step", round(d["ms_per_step"], 2), "its", d["config"]["iterations"], "transport", d["config"].get("transport"), "scaling", d["scaling"])
    except Exception as e:
        print(n, "no line:", e)
PY
