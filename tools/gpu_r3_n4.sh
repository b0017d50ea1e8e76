#!/bin/bash
# N4 slice and the sharded tests on the MI355X: GPU tests of the device CSR algebra and of the reference's
# MergedOperator.parse on the HIP library, then Div @ Flux on the headline grid: device vs scipy on the host.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r3n}
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || true
timeout 900 python -m pytest tests -m gpu -q -x -k "device_csr or merged_operator" > $O/pytest_n4.log 2>&1
echo "pytest exit $?"; tail -3 $O/pytest_n4.log
timeout 600 python tools/csr_algebra_bench.py 40 > $O/csr_bench_n40.json 2> $O/csr_bench_n40.err; echo "n40 exit $?"; cat $O/csr_bench_n40.json; tail -3 $O/csr_bench_n40.err
timeout 900 python tools/csr_algebra_bench.py 69 > $O/csr_bench_n69.json 2> $O/csr_bench_n69.err; echo "n69 exit $?"; cat $O/csr_bench_n69.json; tail -3 $O/csr_bench_n69.err
