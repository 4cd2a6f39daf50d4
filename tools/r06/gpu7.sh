# round 6, call 7: kernel table of the configs[4] line (hd = 128 flash prefill kernel in place)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rm -rf /tmp/kt; timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-strong --no-pmc --lm opt67 --shots 32 --lm-weights fp8_mfma > $O/r06_opt67fp8_prof_bench.log 2>&1
python $R/tools/rocpd_stats.py /tmp/kt/kt_results.db > $O/r06_opt67fp8_kernel_stats.md 2>&1
head -30 $O/r06_opt67fp8_kernel_stats.md | cut -c1-200
