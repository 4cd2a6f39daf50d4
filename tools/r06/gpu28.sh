# round 6, call 28: timelines on the PRODUCT's 16 x 16 instances (the trace used to force the 32 x 32 ones)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
for s in fc2_st proj_st; do timeout 300 python $R/tools/gemm_timeline.py $s 279616 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | head -7; done > $O/r06_gemm_timeline_m16.log
TRACE_FLAGS=33816576 timeout 300 python $R/tools/gemm_timeline.py fc2_st 279616 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | head -20 >> $O/r06_gemm_timeline_m16.log
cat $O/r06_gemm_timeline_m16.log
