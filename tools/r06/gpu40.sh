# round 6, call 40: ViT fc2 (statistics producer, N = 1408, K = 6144): three-deep A ring x dynamic per-XCD tile scheduler
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
C=$R/eilev_amd/csrc
{
echo "== A = two step buffers, B = three-deep A ring (both static stride)"
AB_SHAPES=fc2_st,fc2,proj_st timeout 600 python $R/tools/gemm_ab.py $C/libeilev_hip_probes_noa3.so $C/libeilev_hip_probes.so 279616 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
echo "== three-deep A ring: static stride (0) vs dynamic per-XCD scheduler (262144)"
PROBE_M=279616 timeout 600 python $R/tools/gemm_probe.py 0,262144 fc2_st,fc2,proj_st,fc1_ln,qkv_ln 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
echo "== two step buffers: static (0) vs dynamic (262144)"
cp $C/libeilev_hip_probes.so /tmp/keep.so; cp $C/libeilev_hip_probes_noa3.so $C/libeilev_hip_probes.so
PROBE_M=279616 timeout 600 python $R/tools/gemm_probe.py 0,262144 fc2_st 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
cp /tmp/keep.so $C/libeilev_hip_probes.so
} > $O/r06_a3_x_dynamic.log 2>&1
cat $O/r06_a3_x_dynamic.log
