# round 6, call 32: A operand in a ring of three K-steps (staging overlays slot 2) vs two step buffers; two builds in one process, product instances, flags 0
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
C=$R/eilev_amd/csrc
{
AB_SHAPES=fc1_ln,qkv_ln,fc2_st,proj_st,fc1_noact,fc2 timeout 900 python $R/tools/gemm_ab.py $C/libeilev_hip_a2.so $C/libeilev_hip.so 279616 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
AB_SHAPES=fc1_ln,fc2_st,proj_st timeout 300 python $R/tools/gemm_ab.py $C/libeilev_hip_a2.so $C/libeilev_hip.so 34952 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
AB_SHAPES=opt_qkv,opt_fc1,opt_fc2,opt_out timeout 300 python $R/tools/gemm_ab.py $C/libeilev_hip_a2.so $C/libeilev_hip.so 279616 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
} > $O/r06_a3_ring_ab.log 2>&1
cat $O/r06_a3_ring_ab.log
