# round 6, call 25: does the K-lock-step probe at least cut the fabric traffic of fc2?  FETCH_SIZE / L2 hits with and without it
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
{
for f in 0 262144; do
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/pm; PROBE_M=279616 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pm -o pm -- python $R/tools/gemm_probe.py $f fc2_st 1 > /dev/null 2>&1
    echo "== fc2_st flags $f :: $grp"; python $R/tools/rocpd_pmc.py /tmp/pm/pm_results.db gemm_pp4 2>&1 | tail -5
  done
done
} > $O/r06_xcd_lockstep_traffic.log 2>&1
cat $O/r06_xcd_lockstep_traffic.log
