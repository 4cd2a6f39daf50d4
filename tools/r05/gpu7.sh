cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
# where does the one-wave-per-SIMD kernel (w6, 12 << 4) stop beating the persistent ping-pong kernel (9 << 4)?  rows sweep, N = 2048 / 2560
for m in 7680 11520 15360 19200 23040 26880 30720; do
  PROBE_NOBIAS=1 PROBE_MT5=$m PROBE_MOPT=$m timeout 300 python $R/tools/gemm_probe.py 144,192 t5_o,t5_wo,opt_out,opt_fc2 3 2>&1 | grep "TF/s"
done > $O/r05_w6_vs_pp4_rows.log 2>&1
cat $O/r05_w6_vs_pp4_rows.log
