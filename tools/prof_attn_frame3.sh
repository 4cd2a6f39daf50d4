#!/bin/bash
# PMC summary of the two frame-attention kernels at 1088 frames (rocprofv3 --kernel-trace --pmc, separate passes per counter group)
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
{
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE"; do
    rm -rf /tmp/pm
    MODES=32 REPS=3 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pm -o pm -- python $R/tools/attn_frame3_probe.py 1088 > /dev/null 2>&1
    echo "== $grp"
    python $R/tools/rocpd_pmc.py /tmp/pm/pm_results.db attn_frame 2>&1 | tail -14
done
} > $O/${TAG:-r04}_attn_frame3_pmc.txt 2>&1
