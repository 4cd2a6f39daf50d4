# per-kernel durations of latency mode (one 16-shot sample: 136-frame ViT launch, L = 960 prefill, batch-1 decode)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf /tmp/kl; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kl -o kl -- python $R/bench.py --samples 1 --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-strong --no-pmc > $O/r04_prof_latency.log 2>&1
python $R/tools/rocpd_stats.py /tmp/kl/kl_results.db > $O/r04_latency_kernel_stats.md 2>&1
head -26 $O/r04_latency_kernel_stats.md | cut -c1-150
