# round 6, call 19: the N > 1 code path of bench.py on one GPU (--share-gpu: ranks over gloo, the exchange through torch.distributed): world 2 and 8, 1 step
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for n in 2 8; do
  timeout 900 python bench.py --gpus $n --share-gpu --steps 1 --warmup 1 --samples 4 --no-strong 2> gpurun_out/r06_share_gpu_$n.err | tail -1 | cut -c1-1200 > gpurun_out/r06_share_gpu_$n.json
  echo "world $n rc=$?"; cut -c1-700 gpurun_out/r06_share_gpu_$n.json; tail -3 gpurun_out/r06_share_gpu_$n.err | cut -c1-300
done
