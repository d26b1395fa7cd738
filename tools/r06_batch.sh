cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06p; mkdir -p $O
(time python -m pytest tests -m gpu -q -x 2>&1 | tail -12) > $O/gputest.log 2>&1
tail -6 $O/gputest.log
