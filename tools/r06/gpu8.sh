# round 6, call 8: timelines of the persistent GEMM at the bench launch size: fc2 (5.5 column tiles), fc2 at N = 1536 / 1280 (whole tiles only), proj
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
for s in fc2_st fc2_n1536 fc2_n1280 proj_st; do timeout 300 python $R/tools/gemm_timeline.py $s 279616 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"; done > $O/r06_gemm_timeline.log
cat $O/r06_gemm_timeline.log
