# round 6, call 1: (a) VALU-under-MFMA probe; (b) the default bench at the round's start (box baseline)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
timeout 300 $R/tools/probes/valu_shadow.bin > $O/r06_valu_shadow.log 2>&1
cat $O/r06_valu_shadow.log
cd $R && timeout 600 python bench.py 2> $O/r06_bench_start.err | tail -1 > $O/r06_bench_start.json
cut -c1-1500 $O/r06_bench_start.json
