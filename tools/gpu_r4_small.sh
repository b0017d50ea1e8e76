#!/bin/bash
# shares of the headline grid on one MI355X: the basis of the strong-scaling budget -> gpurun_out/r4small/small_steps.log
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
O=gpurun_out/r4small; mkdir -p $O
echo "# bench.py --no-cpu-baseline --no-extra-configs --steps 5 --n-side N [--force-sharded]: shares of the headline grid on one MI355X (round 4, final tree)" > $O/small_steps.log
for v in "n35:--n-side 35" "n35_sharded:--n-side 35 --force-sharded" "n44:--n-side 44" "n55:--n-side 55" "n69_sharded:--force-sharded"; do
  n="${v%%:*}"; a="${v#*:}"
  timeout 400 python bench.py --no-cpu-baseline --no-extra-configs --steps 5 $a > $O/$n.json 2> $O/$n.err
  python - "$O" "$n" >> $O/small_steps.log <<'PY'
import json, sys
o, n = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"{o}/{n}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"{n + '.json':20s} ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} cells {d['config']['cells_per_gpu']} amg_setup {d['config']['amg']['setup_ms']:.2f} {ph} transport {d['config'].get('transport')}")
except Exception as e:
    print(n, "FAILED", e, open(f"{o}/{n}.err").read()[-600:])
PY
done
cat $O/small_steps.log
