# Round-4 profiling pass (run on the GPU box through gpurun; results land in gpurun_out/ and are copied into profiles/).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
# 1) kernel trace + stats of the bench command (no CPU legs: they are not kernels)
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-pmc > $O/r04_prof_bench.log 2>&1
python $R/tools/rocpd_stats.py /tmp/kt/kt_results.db > $O/r04_bench_kernel_stats.md 2>&1
# 2) PMC counters of the four ViT GEMM shapes at the bench launch shape, one counter group per pass (--kernel-trace only)
for sh in fc1_ln qkv_ln fc2_st proj_st; do
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/pm; PROBE_M=279616 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pm -o pm -- python $R/tools/gemm_probe.py 0 $sh 1 > /dev/null 2>&1
    echo "== $sh :: $grp"; python $R/tools/rocpd_pmc.py /tmp/pm/pm_results.db gemm_pp4 2>&1 | tail -12
  done
done > $O/r04_gemm_pmc.txt 2>&1
tail -2 $O/r04_prof_bench.log | cut -c1-400
# 3) the vendor GEMM next to ours on the same box (bias-only epilogues, bench launch shapes)
python $R/tools/blas_yardstick.py > $O/r04_blas_yardstick.log 2>&1
tail -6 $O/r04_blas_yardstick.log
