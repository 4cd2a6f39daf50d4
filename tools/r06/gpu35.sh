# round 6, call 35: T5 / real-shape / stage tests with the K >= 5120 rule, and the configs[3] line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_t5.py tests/test_hip_full_depth_t5.py tests/test_hip_real_shapes.py tests/test_hip_stages.py tests/test_hip_kernels.py tests/test_ln_fold.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-strong --no-pmc --lm t5xl 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phases_rank0'], d['lm_phase'])"
