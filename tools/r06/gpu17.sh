# round 6, call 17: the whole GPU suite at HEAD without -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r06_gputest_summary.log
cat gpurun_out/r06_gputest_summary.log | grep -v Warning | tail -30
