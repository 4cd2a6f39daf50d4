# round 6, call 9: the whole-K-step half tile (HT16): correctness at the bench shapes, A/B against the four-barrier half tile (probe flag 65536), timeline
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_real_shapes.py tests/test_ln_fold.py -q -m gpu -x -k "bench_shape_gemm or vit_bench_launch or ln_fold or folded or head_major" 2>&1 | tail -4
cd /tmp
PROBE_M=279616 timeout 900 python $R/tools/gemm_probe.py 0,65536 fc2_st,proj_st,fc2,proj 7 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" > $O/r06_ht16_ab.log
PROBE_M=34952 timeout 300 python $R/tools/gemm_probe.py 0,65536 fc2_st,proj_st 7 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" >> $O/r06_ht16_ab.log
cat $O/r06_ht16_ab.log
for s in fc2_st proj_st; do timeout 300 python $R/tools/gemm_timeline.py $s 279616 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | head -6; done > $O/r06_gemm_timeline_ht16.log
cat $O/r06_gemm_timeline_ht16.log
