#!/bin/bash
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
ENVF=$(python - <<'PY'
import oracle
e = oracle.ref_env(extra_last=["."], prefer_archive=True)
print(e["PYTHONPATH"] if e else "")
PY
)
cd /tmp
F="PFV_FUZZ_DEVICE=1 PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=$ENVF:$R"
for spec in "300122:2,6" "400077:6,10" "500049:10,15" "500091:10,15"; do
  seed=${spec%%:*}; dec=${spec#*:}
  for v in "default:" "noredowide:PFV_MPSA_REDO_WIDE=0" "nodd:PFV_MPSA_DD=0"; do
    name=${v%%:*}; envs=${v#*:}
    echo "== seed $seed $name"
    env $F $envs PFV_FUZZ_DECADES=$dec timeout 600 python $R/tools/fuzz_vs_reference.py 1 $seed contrast 2>&1 | grep -i "mech" | cut -c1-330
  done
done
