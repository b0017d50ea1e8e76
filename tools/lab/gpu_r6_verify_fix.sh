export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_mpsa.py -m gpu -q --timeout 600 -k "singular_corner or contrast or replays or golden" 2>&1 | tail -3
bash tools/lab/gpu_r6_bisect.sh 2>&1 | grep "==\|suspicious\|core\|fault" | head -20
