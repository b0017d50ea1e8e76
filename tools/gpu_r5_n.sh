#!/bin/bash
# Round 5, late: member lists kept with the kept aggregate maps (PFV_AMG_MEMBERS_REUSE) -- A/B + the solver tests
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5n
mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra-configs --no-whole-grid-check > $O/$tag.json 2> $O/$tag.err
  python - "$O/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:20s} ms/step {d['ms_per_step']:.2f} cold {d['ms_per_step_cold']:.2f} its {d['config']['iterations']} solve {d['assembly']['phases_ms']['solve_ms']:.2f} amg_setup {d['config']['amg']['setup_ms']:.2f} setup launches {d['solve_launches']['amg_setup']} resid {d['config'].get('true_rel_residual')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
}
run direct PFV_AMG_GALERKIN_DIRECT=1
run passes PFV_AMG_GALERKIN_DIRECT=0
run members_rebuilt PFV_AMG_MEMBERS_REUSE=0
run direct2 PFV_AMG_GALERKIN_DIRECT=1
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-whole-grid-check --no-cold > $O/with_configs.json 2> $O/with_configs.err
python - "$O/with_configs.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("with configs: ms/step", round(d["ms_per_step"], 2), "c2", round(d["config_c2"]["ms_per_step"], 2), "c4", round(d["config_c4"]["ms_per_step"], 2), d["config_c4"]["phases_ms"], "its", d["config_c4"]["iterations"])
except Exception as e:
    print("with configs FAILED", e, open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
for mr in 1; do
  PFV_AMG_MEMBERS_REUSE=$mr timeout 300 python bench.py --force-sharded --steps 6 --warmup 3 --no-cpu-baseline --no-extra-configs --no-whole-grid-check --no-cold > $O/sharded_$mr.json 2> $O/sharded_$mr.err
  python - "$O/sharded_$mr.json" "sharded members_reuse=$mr" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:28s} ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} phases {d['assembly']['phases_ms']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
done
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "amg or solve or block or shard or csr or golden_case or headline or dropin" > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
