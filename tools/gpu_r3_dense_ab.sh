export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3u; mkdir -p $O
for v in lds1:PFV_AMG_DENSE_LDS=1 lds0:PFV_AMG_DENSE_LDS=0 lds1b:PFV_AMG_DENSE_LDS=1; do
  n="${v%%:*}"; e="${v#*:}"
  env $e timeout 400 python bench.py --no-cpu-baseline --steps 6 > $O/ab_$n.json 2> $O/ab_$n.err
  python - "$O" "$n" <<'PY'
import json, sys
o, n = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"{o}/ab_{n}.json") if l.startswith("{")][-1])
    print(f"{n:8s} ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} amg_setup {d['config']['amg']['setup_ms']:.2f} "
          f"c2 {d['config_c2']['ms_per_step']:.2f} (solve {d['config_c2']['phases_ms']['solve_ms']:.2f}) "
          f"c4 {d['config_c4']['ms_per_step']:.1f} node {d['config_c4']['phases_ms']['node_ms']:.1f} its {d['config_c4']['iterations']}")
except Exception as e:
    print(n, "FAILED", e, open(f"{o}/ab_{n}.err").read()[-600:])
PY
done
