# round 6, call 29: half-tile pairing (262144) / rotation of the deal (33554432) against the plain deal (0 at this size; 65536 forces it) ON THE PRODUCT'S 16 x 16 INSTANCES
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
{
PROBE_M=279616 timeout 900 python $R/tools/gemm_probe.py 0,262144,33554432 fc2_st,proj_st,fc2,proj 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
PROBE_M=34952 timeout 300 python $R/tools/gemm_probe.py 65536,262144,33619968 fc2_st,proj_st 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
TRACE_FLAGS=33554432 timeout 300 python $R/tools/gemm_timeline.py fc2_st 279616 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | head -6
} > $O/r06_deal_on_m16_ab.log 2>&1
cat $O/r06_deal_on_m16_ab.log
