cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
C=$R/eilev_amd/csrc
export AB_SHAPES=fc1_noact,fc2,qkv_ln
for B in sk1 sk2 sk3 sk12; do
echo "== A = libeilev_hip_wb.so, B = libeilev_hip_$B.so (timing probe: B skips LDS-DMA pieces, its results are wrong by construction)"
timeout 900 python $R/tools/gemm_ab.py $C/libeilev_hip_wb.so $C/libeilev_hip_$B.so 279616 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
done > $O/r05_dma_skip_probe.log 2>&1
cat $O/r05_dma_skip_probe.log
