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
for v in "default:" "nopos:PFV_ASM_POS=0" "noredowide:PFV_MPSA_REDO_WIDE=0" "nodd:PFV_MPSA_DD=0" "nosymb:PFV_SYMB_REUSE=0"; do
  name=${v%%:*}; envs=${v#*:}
  echo "== $name"
  env $F $envs PFV_FUZZ_DECADES=2,6 timeout 300 python $R/tools/fuzz_vs_reference.py 6 300000 contrast 2>&1 | tail -3 | cut -c1-200
done
