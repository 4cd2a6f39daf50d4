# round 6, call 23: K-lock-step of an XCD's workgroups in the persistent GEMM (probe flag 262144): split-phase barrier every 8 K-steps, 4 of slack
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
PROBE_M=279616 timeout 600 python $R/tools/gemm_probe.py 0,262144 fc2_st,proj_st,fc1_ln,qkv_ln 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" > $O/r06_xcd_lockstep_ab.log
cat $O/r06_xcd_lockstep_ab.log
