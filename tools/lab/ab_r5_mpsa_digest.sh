export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in porepy_amd/csrc/libporefv_hip.so tools/lab/libporefv_hip_r5.so; do
python - "$lib" <<'PY' 2>&1 | tail -2
import sys, os
sys.path.insert(0, ".")
import porepy_amd as pa
from tests import _parity as P
lib = pa._lib.load_library(os.path.abspath(sys.argv[1]))
out = P.mpsa_whole_grid_check(lib, 44)
print(sys.argv[1], {k: [float(f"{x:.2e}") for x in out[k]] for k in P.MPSA_KEYS})
PY
done
