cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
C=$R/eilev_amd/csrc
A=${A:-head}; B=${B:-wb}
export AB_SHAPES=${AB_SHAPES:-fc1_ln,qkv_ln,fc2_st,proj_st,fc2,fc1_noact}
echo "== A = libeilev_hip_$A.so, B = libeilev_hip_$B.so" > $O/r05_ab_${A}_${B}.log
timeout 900 python $R/tools/gemm_ab.py $C/libeilev_hip_$A.so $C/libeilev_hip_$B.so 279616 7 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" >> $O/r05_ab_${A}_${B}.log
cat $O/r05_ab_${A}_${B}.log
