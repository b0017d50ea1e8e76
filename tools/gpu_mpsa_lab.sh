#!/bin/bash
# stage times of the MPSA interaction-region kernel: bash tools/gpu_mpsa_lab.sh [extra env ...]
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for k in 1 2 3 4 5 0; do
  env "$@" PFV_MPSA_ABLATE=$k timeout 300 python tools/mpsa_lab.py 2>&1 | grep -v amdgpu.ids
done
