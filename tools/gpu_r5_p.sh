#!/bin/bash
# Round 5, late: the sharded path on one rank -- windows of the owned rows handed to the AMG setup, child hierarchy follows the parent's kept maps
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5p
mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --force-sharded --steps 8 --warmup 3 --no-cpu-baseline --no-extra-configs --no-whole-grid-check --no-cold > $O/$tag.json 2> $O/$tag.err
  python - "$O/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"{sys.argv[2]:22s} ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} amg_setup {d['config'].get('amg', {}).get('setup_ms')} {ph}")
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
}
run sharded_new PFV_X=0
run sharded_no_child PFV_AMG_REUSE_CHILD=0
run sharded_no_win0 PFV_SHARD_WIN0=0
run sharded_old PFV_AMG_REUSE_CHILD=0 PFV_SHARD_WIN0=0
run sharded_new2 PFV_X=0
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra-configs --no-whole-grid-check --no-cold > $O/single.json 2> $O/single.err
python - "$O/single.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("single path            ms/step", round(d["ms_per_step"], 2), "its", d["config"]["iterations"])
PY
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -k "shard or amg or solve" > $O/pytest_sel.log 2>&1; tail -2 $O/pytest_sel.log
