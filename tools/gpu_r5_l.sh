#!/bin/bash
# Round 5, call L: the sharded path on one rank (RCCL) with the permeability resident in HBM; reuse of the coupled hierarchy's maps A/B.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5l
mkdir -p $O
run() {  # tag, args, env...
  tag=$1; shift; args=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --no-whole-grid-check --no-cold $args > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
o, tag = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"{o}/bench_{tag}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"{tag}: ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} cells {d['config']['cells_per_gpu']} asm {d['assembly']['ms']:.2f} {ph} amg_setup {d['config']['amg']['setup_ms']:.2f} reused {d['config']['pattern_reuse']['amg_aggregate_maps_kept']} transport {d['config'].get('transport')}")
except Exception as e:
    print(tag, "bench FAILED", e, open(f"{o}/bench_{tag}.err").read()[-1500:])
PY
}
run n69_sharded "--steps 6 --force-sharded" PFV_X=0
run n69_sharded_noreuse "--steps 6 --force-sharded" PFV_AMG_REUSE_DIST=0
run n69_single "--steps 6" PFV_X=0
run n35_sharded "--steps 8 --n-side 35 --force-sharded" PFV_X=0
run n35_sharded_noreuse "--steps 8 --n-side 35 --force-sharded" PFV_AMG_REUSE_DIST=0
run n35_single "--steps 8 --n-side 35" PFV_X=0
run n35_single_noreuse "--steps 8 --n-side 35" PFV_AMG_REUSE_REBUILT=0
