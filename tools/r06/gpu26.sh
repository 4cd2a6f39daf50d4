# round 6, call 26: K-lock-step probe on the product's 16 x 16 instances (flag 262144) against free-running (0); 65536 = any other flag = the 32 x 32 instance
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
PROBE_M=279616 timeout 600 python $R/tools/gemm_probe.py 0,262144,65536 fc2_st,proj_st,fc1_ln,qkv_ln 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" > $O/r06_xcd_lockstep_ab2.log
cat $O/r06_xcd_lockstep_ab2.log
