#!/bin/bash
# Round 6, checkpoint C: one handle beyond 6.3 M cells (implicit vector_source pattern): the step at n_side 104 / 110,
# rows against the split path (tools/big_handle_check.py); the two new GPU tests.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6c
mkdir -p $O
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $O/timeline.log; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "flux_pattern or pipeline or rebuilt_topology" > $O/pytest_subset.log 2>&1
echo "pytest exit $?" >> $O/pytest_subset.log; tail -4 $O/pytest_subset.log; stamp pytest
timeout 1500 python tools/big_handle_check.py 110 24 > $O/big_handle_110.json 2> $O/big_handle_110.err
tail -c 1500 $O/big_handle_110.json; tail -c 600 $O/big_handle_110.err; stamp big_handle
for n in 104 110; do
  timeout 1200 python bench.py --n-side $n --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --no-whole-grid-check > $O/bench_$n.json 2> $O/bench_$n.err
  python - "$O" "$n" <<'PY'
import json, sys
o, n = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"{o}/bench_{n}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"n_side {n}: cells {d['config']['global_cells']} ms/step {d['ms_per_step']:.1f} cold {d['ms_per_step_cold']:.1f} value {d['value']:.3e} its {d['config']['iterations']} asm frac {d['assembly']['frac_of_hbm_peak']:.3f} {ph} roofline {d['roofline']['frac']:.3f}")
except Exception as e:
    print(n, "FAILED", e, open(f"{o}/bench_{n}.err").read()[-1200:])
PY
done
stamp bench
cat $O/timeline.log
