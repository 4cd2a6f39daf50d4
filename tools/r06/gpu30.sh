cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
for s in fc2_st proj_st fc1_noact; do timeout 300 python $R/tools/gemm_timeline.py $s 279616 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | head -15; done > $O/r06_gemm_timeline_m16_xcd.log
cat $O/r06_gemm_timeline_m16_xcd.log
