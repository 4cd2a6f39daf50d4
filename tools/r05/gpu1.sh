# round 5, call 1: (a) early-arrive A/B of the ping-pong GEMM; (b) configs[3] / configs[4] lines with kernel tables
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
C=$R/eilev_amd/csrc
export AB_SHAPES=fc1_ln,qkv_ln,fc2_st,proj_st,fc2,fc1_noact
for v in "" _e14 _e12; do
  echo "== A = HEAD, B = libeilev_hip$v.so"
  timeout 600 python $R/tools/gemm_ab.py $C/libeilev_hip_head.so $C/libeilev_hip$v.so 279616 7 2>&1 | grep -v "^W2026\|^E2026"
done > $O/r05_early_ab.log 2>&1
tail -30 $O/r05_early_ab.log
for cfg in "t5xl:--lm t5xl" "opt67fp8:--lm opt67 --shots 32 --lm-weights fp8_mfma"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-strong --no-pmc $fl > $O/r05_${tag}_prof_bench.log 2>&1
  python $R/tools/rocpd_stats.py /tmp/kt/kt_results.db > $O/r05_${tag}_kernel_stats.md 2>&1
  grep -v "^W2026\|^E2026" $O/r05_${tag}_prof_bench.log | tail -1 | cut -c1-900
done
