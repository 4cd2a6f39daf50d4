cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for i in 1 2 3; do timeout 900 python -m pytest tests/test_hip_sharded.py -q -m gpu > /tmp/sh_$i.log 2>&1; grep -E "passed|failed|sharded_check failed" /tmp/sh_$i.log | cut -c1-300; done > gpurun_out/r06_sharded_repeat.log 2>&1
cat gpurun_out/r06_sharded_repeat.log
