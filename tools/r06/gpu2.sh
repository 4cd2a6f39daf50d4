# round 6, call 2: half-tile pairing A/B inside the probe library (flag 65536 = the old deal), bench-size launches
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
PROBE_M=279616 timeout 900 python $R/tools/gemm_probe.py 0,65536 fc2_st,proj_st,fc2,proj 7 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" > $O/r06_pair_ab.log
cat $O/r06_pair_ab.log
PROBE_M=34952 timeout 300 python $R/tools/gemm_probe.py 0,65536 fc2_st,proj_st 7 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" > $O/r06_pair_ab_136.log
cat $O/r06_pair_ab_136.log
