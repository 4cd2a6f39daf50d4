# Round-6 evidence pass (run on the GPU box through gpurun; results land in gpurun_out/ and are copied into profiles/).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
# 1) kernel trace + stats of the bench command (no CPU legs: they are not kernels); the bench line of the SAME run
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-strong --no-pmc > $O/r06_prof_bench.log 2>&1
python $R/tools/rocpd_stats.py /tmp/kt/kt_results.db > $O/r06_final_bench_kernel_stats.md 2>&1
grep -v "^W2026\|^E2026" $O/r06_prof_bench.log | tail -1 > $O/r06_final_bench_under_rocprof.json
# 2) PMC counters of the four ViT GEMM shapes at the bench launch shape, one counter group per pass (--kernel-trace only)
for sh in fc1_ln qkv_ln fc2_st proj_st; do
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/pm; EILEV_PROBE_ON_PRODUCT_LIB=1 PROBE_M=279616 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pm -o pm -- python $R/tools/gemm_probe.py 0 $sh 1 > /dev/null 2>&1
    echo "== $sh :: $grp"; python $R/tools/rocpd_pmc.py /tmp/pm/pm_results.db gemm_pp4 2>&1 | tail -12
  done
done > $O/r06_gemm_pmc.txt 2>&1
python $R/tools/pmc_to_traffic.py $O/r06_gemm_pmc.txt > $O/r06_gemm_traffic.json 2>/dev/null
# 3) the vendor GEMM next to ours on the same box (bias-only epilogues, bench launch shapes)
python $R/tools/blas_yardstick.py > $O/r06_blas_yardstick.log 2>&1
tail -8 $O/r06_blas_yardstick.log
# 4) the default bench command (live traffic measurement, CPU baseline, verification) and the other configs
cd $R
timeout 1800 python bench.py --cpu-full 2> $O/r06_final_bench.err | tail -1 > $O/r06_final_bench.json
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-strong --no-pmc --lm t5xl 2>/dev/null | tail -1 > $O/r06_t5xl_bench_final.json
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-strong --no-pmc --lm opt67 --shots 32 --lm-weights fp8_mfma 2>/dev/null | tail -1 > $O/r06_opt67fp8_bench_final.json
timeout 600 python bench.py --samples 1 --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-strong --no-pmc 2>/dev/null | tail -1 > $O/r06_bench_latency.json
for f in r06_final_bench r06_t5xl_bench_final r06_opt67fp8_bench_final r06_bench_latency; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['value'], d['ms_per_step'], d['phases_rank0'])"; done
