# round 6, call 39: timelines of plain fc2 (N = 1408, K = 6144) with and without the three-deep A ring: what do half tiles cost under it?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
C=$R/eilev_amd/csrc
{
echo "== three-deep A ring (product rule)"; timeout 300 python $R/tools/gemm_timeline.py fc2 279616 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | head -6
cp $C/libeilev_hip_probes.so /tmp/probes_keep.so; cp $C/libeilev_hip_probes_noa3.so $C/libeilev_hip_probes.so
echo "== two step buffers"; timeout 300 python $R/tools/gemm_timeline.py fc2 279616 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | head -6
cp /tmp/probes_keep.so $C/libeilev_hip_probes.so
} > $O/r06_a3_halftile_timeline.log 2>&1
cat $O/r06_a3_halftile_timeline.log
