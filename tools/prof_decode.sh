# Round-4: per-kernel durations of the decode step at the given batch sizes (rocprofv3 kernel trace of tools/decode_probe.py).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
TAG=${TAG:-r04}
for b in ${BATCHES:-1 32}; do
  rm -rf /tmp/kd$b; PROBE_B=$b PROBE_FLAG=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kd$b -o kd -- python $R/tools/decode_probe.py > $O/${TAG}_decode_probe_b$b.log 2>&1
  python $R/tools/rocpd_stats.py /tmp/kd$b/kd_results.db > $O/${TAG}_decode_b${b}_kernel_stats.md 2>&1
  grep "^round" $O/${TAG}_decode_probe_b$b.log | tail -2
  head -14 $O/${TAG}_decode_b${b}_kernel_stats.md | tail -8 | cut -c1-150
done
