cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
{
echo "hd = 128 flash prefill (OPT-6.7B, L = 1872): 8 waves, one 32-key block at a time (no spills) | ATTN_DBG=64: 4 waves, two blocks"
python $R/tools/attn_prefill_probe.py opt67 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
ATTN_DBG=64 python $R/tools/attn_prefill_probe.py opt67 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
cd $R && timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "attention" -x 2>&1 | tail -2
} > $O/r06_attn_hd128_variants.log 2>&1
cat $O/r06_attn_hd128_variants.log
