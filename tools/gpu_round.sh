#!/bin/bash
# One gpurun call: GPU parity suite, bench lines (default / sharded drivers / two ranks sharing the GPU
# over gloo) and a kernel trace.  Most important first: the call may be cut.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
mkdir -p $O
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $O/timeline.log; }
stamp start
timeout 600 python -m pytest tests -m gpu -q --timeout 180 --durations=12 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.log
stamp pytest
timeout 330 python bench.py > $O/bench_default.json 2> $O/bench_default.err
stamp bench_default
timeout 200 python bench.py --force-sharded --no-cpu-baseline --no-extra-configs > $O/bench_sharded_library.json 2> $O/bench_sharded_library.err
stamp bench_sharded_library
PFV_BENCH_SHARE_GPU=1 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29517 bench.py --gpus 2 --n-side 40 --no-cpu-baseline --no-extra-configs > $O/bench_2rank_shared_gpu.json 2> $O/bench_2rank_shared_gpu.err
stamp bench_2rank
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -- python $OLDPWD/bench.py --no-cpu-baseline --no-extra-configs > $OLDPWD/$O/bench_traced.json 2> $OLDPWD/$O/rocprof.err)
stamp rocprof
for db in $(find $O/prof -name '*.db' | head -1); do python profiles/summarize_rocpd.py $db > $O/kernel_stats.txt 2>> $O/rocprof.err; done
find $O/prof -name '*.db' -size +40M -delete
PFV_SHARDED_DRIVER=torch timeout 200 python bench.py --force-sharded --no-cpu-baseline --no-extra-configs > $O/bench_sharded_torch.json 2> $O/bench_sharded_torch.err
stamp bench_sharded_torch
tail -3 $O/pytest_gpu.log
cat $O/timeline.log
