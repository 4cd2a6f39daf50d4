cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8; do timeout 600 python -m pytest tests/test_hip_sharded.py -q -m gpu -k "8-1" > /tmp/sh_$i.log 2>&1; grep -E "passed|failed|sharded_check failed" /tmp/sh_$i.log | cut -c1-260; done > gpurun_out/r06_sharded_repeat8.log 2>&1
cat gpurun_out/r06_sharded_repeat8.log
