cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests/test_hip_real_shapes.py tests/test_hip_full_depth_c2.py tests/test_hip_sharded.py -q -m gpu -x 2>&1 | grep -E "passed|failed"
