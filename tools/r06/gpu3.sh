# round 6, call 3: half-tile pairing by launch size (262144 = pairing forced, 65536 = old deal), rotation of the deal (33554432 on top of the old deal)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
for m in 34952 69904 139808 279616; do
PROBE_M=$m timeout 900 python $R/tools/gemm_probe.py 65536,262144,33619968 fc2_st,proj_st 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids"
done > $O/r06_pair_rot_ab.log
cat $O/r06_pair_rot_ab.log
