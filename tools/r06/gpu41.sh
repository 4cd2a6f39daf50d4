# round 6, call 41: three-deep A ring x dynamic per-XCD scheduler (mailbox in memory), ViT GEMMs
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
{
PROBE_M=279616 timeout 600 python $R/tools/gemm_probe.py 0,262144 fc2_st,fc2,proj_st,fc1_ln,qkv_ln 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
} > $O/r06_a3_x_dynamic2.log 2>&1
cat $O/r06_a3_x_dynamic2.log
