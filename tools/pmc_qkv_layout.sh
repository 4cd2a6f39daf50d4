cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for hm in 0 1; do
  for grp in "WRITE_SIZE FETCH_SIZE" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_WRREQ_STALL_sum" "TCC_HIT_sum TCC_MISS_sum TCC_WRITE_sum" "TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    rm -rf /tmp/pm; HM=$hm timeout 400 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pm -o pm -- python $R/tools/vit_hm_ab.py > /dev/null 2>&1
    echo "== block order $hm :: $grp"; python $R/tools/rocpd_pmc.py /tmp/pm/pm_results.db "gemm_pp4_kernel<0, false, 1, 1>" 2>&1 | tail -6
  done
done
