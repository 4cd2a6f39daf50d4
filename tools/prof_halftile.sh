# PMC evidence for the half-tile diagnosis (DESIGN 3b, r4 late): L2 hit rate and fabric fetch of the fc2 GEMM with 5.5 (N = 1408) and with 5 / 6 whole
# column tiles (N = 1280 / 1536), same kernel, same box.  Run on the GPU box through gpurun; result -> gpurun_out/r04_halftile_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
for sh in fc2 fc2_n1280 fc2_n1536; do
  for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pm; PROBE_M=279616 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pm -o pm -- python $R/tools/gemm_probe.py 0 $sh 1 > /dev/null 2>&1
    echo "== $sh :: $grp"; python $R/tools/rocpd_pmc.py /tmp/pm/pm_results.db gemm_pp4 2>&1 | tail -6
  done
done > $O/r04_halftile_pmc.txt 2>&1
tail -60 $O/r04_halftile_pmc.txt
