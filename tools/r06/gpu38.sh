# round 6, call 38: where does the three-deep A ring pay?  N = 1280 / 1408 / 2560 at K = 6144 (and K = 10240), M = 279616 and 30720; A = no A3 anywhere, B = product (A3 from K >= 5120)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
C=$R/eilev_amd/csrc
{
for m in 279616 30720; do
AB_SHAPES=opt_x1280,opt_x1408,opt_x2560,opt_x2560k10 AB_MOPT=$m timeout 900 python $R/tools/gemm_ab.py $C/libeilev_hip_noa3.so $C/libeilev_hip.so $m 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
done
} > $O/r06_a3_where.log 2>&1
cat $O/r06_a3_where.log
