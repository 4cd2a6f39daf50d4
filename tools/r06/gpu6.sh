# round 6, call 6: hd = 128 flash prefill kernel (tests + configs[4] line), the timed-decode-shape parity test, the ABI-16 fold tests
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "attention" -x 2>&1 | tail -5
timeout 900 python -m pytest tests/test_hip_real_shapes.py -q -m gpu -k "timed_bench_shape or batch_decode_step" -x 2>&1 | tail -5
timeout 900 python -m pytest tests/test_ln_fold.py tests/test_hip_stages.py -q -m gpu -x 2>&1 | tail -5
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-strong --no-pmc --lm opt67 --shots 32 --lm-weights fp8_mfma 2> $O/r06_opt67fp8.err | tail -1 > $O/r06_opt67fp8_bench_hd128.json
cut -c1-2500 $O/r06_opt67fp8_bench_hd128.json
