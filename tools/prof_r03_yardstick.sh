# Round-3 yardstick pass: the vendor bf16 GEMM next to eilev_linear on the ViT launch shapes (same box, same data), kernel-trace durations
# and PMC counters (separate passes) of BOTH, so that the gap is located (MFMA busy, clock, L2 hit, fabric bytes).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
python $R/tools/blas_yardstick.py > $O/r03_blas_yardstick.log 2>&1
for sh in fc1 fc2 qkv proj; do
  rm -rf /tmp/kt; YARD_SHAPES=$sh YARD_ROUNDS=2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/blas_yardstick.py > /dev/null 2>&1
  echo "=== $sh"; python $R/tools/rocpd_stats.py /tmp/kt/kt_results.db 2>&1 | grep -E "Cijk|gemm_|^\|" | head -8
done > $O/r03_yardstick_kernel_stats.md 2>&1
for sh in fc1 fc2; do
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pm; YARD_SHAPES=$sh YARD_ROUNDS=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pm -o pm -- python $R/tools/blas_yardstick.py > /dev/null 2>&1
  echo "== $sh :: $grp"; python $R/tools/rocpd_pmc.py /tmp/pm/pm_results.db 2>&1 | grep -A10 -E "^## (Custom_Cijk|Cijk|gemm_)" 
done; done > $O/r03_yardstick_pmc.txt 2>&1
cat $O/r03_blas_yardstick.log
