cd /tmp && export TMPDIR=/tmp
R=/root/repo
# 1) kernel trace + stats of the bench command
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 > $R/gpurun_out/prof_bench.log 2>&1
python $R/tools/rocpd_stats.py /tmp/kt/kt_results.db > $R/gpurun_out/r01_final_bench_kernel_stats.md 2>&1
# 2) PMC traffic of the four ViT GEMM shapes at the bench launch shape
for sh in fc1 qkv fc2 proj; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm; PROBE_M=279616 timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pm -o pm -- python $R/tools/gemm_probe.py 0 $sh 1 > /dev/null 2>&1
    echo "== $sh $c"; python $R/tools/rocpd_pmc.py /tmp/pm/pm_results.db gemm_ 2>&1 | tail -8
  done
done > $R/gpurun_out/pmc_traffic.txt 2>&1
tail -3 $R/gpurun_out/prof_bench.log | cut -c1-300
