cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
for m in 3840 7680 15360; do
  PROBE_MT5=$m PROBE_MOPT=$m timeout 300 python $R/tools/gemm_probe.py 144,192 opt_qkv,opt_fc1,t5_qkv,t5_wi 3 2>&1 | grep "TF/s"
done > $O/r05_w6_vs_pp4_rows2.log 2>&1
cat $O/r05_w6_vs_pp4_rows2.log
