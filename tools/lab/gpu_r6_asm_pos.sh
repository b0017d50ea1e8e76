export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "rebuilt_topology or golden_case or amg_preconditioner or mid_size" 2>&1 | tail -3
for v in "pos1:" "pos0:PFV_ASM_POS=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout 600 python bench.py --steps 8 --warmup 3 --no-cold --no-cpu-baseline --no-whole-grid-check --no-extra-configs > gpurun_out/q_$name.json 2> gpurun_out/q_$name.err
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.loads([l for l in open(f"gpurun_out/q_{name}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"{name}: ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} asm {d['assembly']['ms']:.2f} frac {d['assembly']['frac_of_hbm_peak']:.3f} {ph}")
except Exception as e:
    print(name, "FAILED", e, open(f"gpurun_out/q_{name}.err").read()[-800:])
PY
done
