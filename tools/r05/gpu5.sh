cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
C=$R/eilev_amd/csrc
export AB_SHAPES=${AB_SHAPES:-fc1_noact,fc2,qkv_ln,fc1_ln,fc2_st,proj_st}
for B in $VARS; do
echo "== A = libeilev_hip_$BASE.so, B = libeilev_hip_$B.so"
timeout 900 python $R/tools/gemm_ab.py $C/libeilev_hip_$BASE.so $C/libeilev_hip_$B.so 279616 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
done > $O/$LOG 2>&1
cat $O/$LOG
