cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_hip_full_depth_c2.py tests/test_hip_full_depth.py tests/test_hip_model_api.py tests/test_hip_kernels.py tests/test_hip_stages.py -x -q -m gpu > $O/r05_gputest_b.log 2>&1; tail -6 $O/r05_gputest_b.log | cut -c1-600
timeout 1200 python bench.py --no-cpu-baseline --no-pmc 2> $O/r05_bench_b.err | tail -1 > $O/r05_bench_b.json
python -c "
import json; d=json.load(open('$O/r05_bench_b.json')); print(d['value'], d['phases_rank0'], d['roofline']['vit_gemm_us_and_tflops']); print(json.dumps(d['reference_parity'])[:3000]); print(d['verified'])"
