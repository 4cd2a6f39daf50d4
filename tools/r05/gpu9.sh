# full GPU tier + smoke + T5 line + default bench line
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r05_gputest_a.log 2>&1; tail -4 $O/r05_gputest_a.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-strong --no-pmc --lm t5xl 2> $O/r05_t5xl_bench_a.err | tail -1 > $O/r05_t5xl_bench_a.json; cut -c1-1500 $O/r05_t5xl_bench_a.json
timeout 1200 python bench.py --no-cpu-baseline --no-pmc 2> $O/r05_bench_a.err | tail -1 > $O/r05_bench_a.json; cut -c1-1200 $O/r05_bench_a.json
