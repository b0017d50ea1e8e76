#!/bin/bash
# Round 6: the randomized differential driver on the MI355X against the reference archive, larger seed counts
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6fuzz
mkdir -p $O
ENVF=$(python - <<'PY'
import oracle
e = oracle.ref_env(extra_last=["."], prefer_archive=True)
print(e["PYTHONPATH"] if e else "")
PY
)
[ -z "$ENVF" ] && { echo "no reference archive"; exit 1; }
cd /tmp
F="PFV_FUZZ_DEVICE=1 PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=$ENVF:$R"
env $F timeout 900 python $R/tools/fuzz_vs_reference.py 300 100000 > $R/$O/plain_300.log 2>&1; tail -1 $R/$O/plain_300.log
env $F timeout 900 python $R/tools/fuzz_vs_reference.py 150 200000 special > $R/$O/special_150.log 2>&1; tail -1 $R/$O/special_150.log
env $F PFV_FUZZ_DECADES=2,6 timeout 600 python $R/tools/fuzz_vs_reference.py 150 300000 contrast > $R/$O/contrast_1e2_1e6_150.log 2>&1; tail -1 $R/$O/contrast_1e2_1e6_150.log
env $F PFV_FUZZ_DECADES=6,10 timeout 1500 python $R/tools/fuzz_vs_reference.py 100 400000 contrast > $R/$O/contrast_1e6_1e10_100.log 2>&1; tail -1 $R/$O/contrast_1e6_1e10_100.log
env $F PFV_FUZZ_DECADES=10,15 timeout 1500 python $R/tools/fuzz_vs_reference.py 100 500000 contrast > $R/$O/contrast_1e10_1e15_100.log 2>&1; tail -1 $R/$O/contrast_1e10_1e15_100.log
