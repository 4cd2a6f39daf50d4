# round 6, call 34: from which K does the three-deep A ring pay?  product (K >= 8192) vs a build with K >= 2048, plain launches at M = 30720
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
C=$R/eilev_amd/csrc
AB_SHAPES=opt_t5_wo,opt_t5_o,opt_67_fc2,opt_67_out,opt_67_qkv,opt_fc2,opt_out,opt_qkv AB_MOPT=30720 timeout 900 python $R/tools/gemm_ab.py $C/libeilev_hip.so $C/libeilev_hip_a3k2.so 30720 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" > $O/r06_a3_min_k.log
cat $O/r06_a3_min_k.log
