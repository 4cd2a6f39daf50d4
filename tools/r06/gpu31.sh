# round 6, call 31: dynamic tile scheduler of the persistent GEMM (probe flag 262144: one counter per XCD; + 65536: one for the chip) vs the static stride, product instances
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
{
PROBE_M=279616 timeout 900 python $R/tools/gemm_probe.py 0,262144,327680 fc1_ln,qkv_ln,fc2_st,proj_st 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
PROBE_M=34952 timeout 300 python $R/tools/gemm_probe.py 0,262144,327680 fc1_ln,fc2_st,proj_st 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
} > $O/r06_dynamic_sched_ab.log 2>&1
cat $O/r06_dynamic_sched_ab.log
