#!/bin/bash
# Round 5, call M: per-step times of the default bench (3 steps) and of a 20-step run (head room of regrowing buffers).
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r5m
mkdir -p $O
for v in "s3:--steps 3 --warmup 1" "s20:--steps 20 --warmup 1" "s20b:--steps 20 --warmup 1"; do
  tag="${v%%:*}"; a="${v#*:}"
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --no-whole-grid-check $a > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
o, tag = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"{o}/bench_{tag}.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"{tag}: ms/step {d['ms_per_step']:.2f} cold {d['ms_per_step_cold']:.2f} prewarm {d['prewarm_steps_untimed']} its {d['config']['iterations']} {ph}")
    print("   each", d["each_timed_step"]["ms"], d["each_timed_step"]["iterations"])
except Exception as e:
    print(tag, "bench FAILED", e, open(f"{o}/bench_{tag}.err").read()[-1500:])
PY
done
