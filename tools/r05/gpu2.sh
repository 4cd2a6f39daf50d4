cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
for sh in fc1_ln fc1_noact qkv_ln fc2_st fc2_n1536; do
  timeout 300 python $R/tools/gemm_itrace.py $R/eilev_amd/csrc/libeilev_hip_itrace.so $sh 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
done > $O/r05_itrace.log 2>&1
cat $O/r05_itrace.log
