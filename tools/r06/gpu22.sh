# round 6, call 22: priority of the MFMA phase in the persistent GEMM: raised per phase (product) vs never raised (prio0) vs static for the late group (prio2)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
C=$R/eilev_amd/csrc
export AB_SHAPES=fc1_ln,qkv_ln,fc2_st,proj_st
for v in prio0 prio2; do
  echo "== A = product (priority raised per MFMA phase), B = $v"
  timeout 900 python $R/tools/gemm_ab.py $C/libeilev_hip.so $C/libeilev_hip_$v.so 279616 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
done > $O/r06_mfma_phase_priority_ab.log 2>&1
cat $O/r06_mfma_phase_priority_ab.log
