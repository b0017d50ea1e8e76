#!/bin/bash
# Round 3, first GPU call: the reference's models on libporefv_hip.so (drop-in tests, product variant), the
# device fuzz against the reference itself, the bench line with the reference-timed cpu_baseline.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3a
mkdir -p $O
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $O/timeline.log; }
stamp start
ls -la oracle/_ref >> $O/timeline.log 2>&1
timeout 900 python -m pytest tests/test_reference_dropin.py -m gpu -q --timeout 600 --durations=8 > $O/pytest_dropin_product.log 2>&1
echo "pytest exit $?" >> $O/pytest_dropin_product.log
stamp dropin
(cd /tmp && PFV_FUZZ_DEVICE=1 PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=$OLDPWD/oracle/shim:$OLDPWD/oracle/_ref/porepy_ref.zip:$OLDPWD timeout 400 python $OLDPWD/tools/fuzz_vs_reference.py 40 9000 > $OLDPWD/$O/fuzz_device_vs_reference.log 2>&1)
stamp fuzz
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
stamp bench_default
tail -5 $O/pytest_dropin_product.log
tail -3 $O/fuzz_device_vs_reference.log
cat $O/timeline.log
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r3a/bench_default.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], json.dumps(d['cpu_baseline'])[:700])
except Exception as e: print('bench parse failed', e)
PY
