# kernel trace + stats of the bench command (no CPU legs: they are not kernels)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${TAG:-r04}
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-strong --no-pmc > $O/${TAG}_prof_bench.log 2>&1
python $R/tools/rocpd_stats.py /tmp/kt/kt_results.db > $O/${TAG}_bench_kernel_stats.md 2>&1
grep -v "^W2026\|^E2026" $O/${TAG}_prof_bench.log | tail -1 | cut -c1-700
