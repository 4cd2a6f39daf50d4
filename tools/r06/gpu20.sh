# round 6, call 20: ViT launch size and the rounds of the static stride: 136 clips per launch (1093 tile rows: fc2 25.6 rounds) vs 138 (1109: 25.99)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for c in 136 138 136 138 142; do
  EILEV_BENCH_CHUNK_CLIPS=$c timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-strong --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chunk $c:', d['value'], d['ms_per_step'], d['phases_rank0']['encode_ms_per_step'], {k:v[1] for k,v in d['roofline']['vit_gemm_us_and_tflops'].items()})"
done > gpurun_out/r06_launch_size_ab.log 2>&1
cat gpurun_out/r06_launch_size_ab.log
