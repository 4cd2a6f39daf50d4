# round 6, call 10: PMC picture of the causal flash prefill kernel (OPT-2.7B shape)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
{
python $R/tools/attn_prefill_probe.py opt27 10 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
python $R/tools/attn_prefill_probe.py opt67 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
python $R/tools/attn_prefill_probe.py t5 10 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES"; do
    rm -rf /tmp/pm
    timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pm -o pm -- python $R/tools/attn_prefill_probe.py opt27 3 > /dev/null 2>&1
    echo "== $grp"
    python $R/tools/rocpd_pmc.py /tmp/pm/pm_results.db attn_prefill 2>&1 | tail -8
done
} > $O/r06_attn_prefill_pmc.txt 2>&1
cat $O/r06_attn_prefill_pmc.txt
