# round 6, call 14: the whole GPU suite at HEAD
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r06_gputest_summary.log
cat gpurun_out/r06_gputest_summary.log
