#!/bin/bash
# Round 5, call H: block copy-out of the symbolic rows (PFV_SYMB_BLOCK_OUT), defaults GJ=5 / MPSA np: A/B + parity.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5h
mkdir -p $O
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --no-whole-grid-check --no-cold --steps 8 > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
o, tag = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"{o}/bench_{tag}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    nk = [k for k in [d["roofline"]] + d["roofline_kernels"] if k["name"] == "node_kernel"][0]
    print(f"{tag}: ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} asm {d['assembly']['ms']:.2f} {ph} node alone {nk['ms_per_launch']:.2f} resid {d['config']['true_rel_residual']:.1e} amg_setup {d['config']['amg']['setup_ms']:.2f}")
except Exception as e:
    print(tag, "bench FAILED", e, open(f"{o}/bench_{tag}.err").read()[-1500:])
PY
}
run default PFV_X=0
run noblockout PFV_SYMB_BLOCK_OUT=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 800 -k "golden_case or generic_pattern or timed_bench_grid or known_answers or config_c2 or partition or batch" > $O/pytest_a.log 2>&1
echo "pytest exit $?" >> $O/pytest_a.log; tail -4 $O/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_mpsa.py -m gpu -q -x --timeout 800 > $O/pytest_mpsa.log 2>&1
echo "pytest exit $?" >> $O/pytest_mpsa.log; tail -4 $O/pytest_mpsa.log
