# round 6, call 33: the three-deep A ring as an own instance for plain launches with K >= 8192 (OPT fc2 of a prefill): tests + the bench's prefill
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_real_shapes.py tests/test_hip_full_depth_c2.py tests/test_hip_full_depth.py tests/test_hip_stages.py tests/test_hip_kernels.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-strong --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phases_rank0'], d['lm_phase'])"
