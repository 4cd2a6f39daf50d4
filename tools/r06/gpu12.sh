cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
{
for d in 0 64 128 66 130; do echo "ATTN_DBG=$d (64: 4 waves per workgroup, 128: 2 waves; +2: no LDS-DMA after the first tile)"; ATTN_DBG=$d python $R/tools/attn_prefill_probe.py opt27 10 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"; done
} > $O/r06_attn_prefill_nwq.log 2>&1
cat $O/r06_attn_prefill_nwq.log
