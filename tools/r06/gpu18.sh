cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_hip_full_depth_t5.py tests/test_hip_t5.py -q -m gpu 2>&1 | tail -3
