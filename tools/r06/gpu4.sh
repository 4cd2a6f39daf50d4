# round 6, call 4: cost of a half tile: one column of half tiles vs one column of whole tiles (4 exact rounds of 256 workgroups)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
PROBE_M=262144 timeout 600 python $R/tools/gemm_probe.py 0 fc2_n128,fc2_n256,proj_n128,proj_n256 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" > $O/r06_halftile_cost.log
cat $O/r06_halftile_cost.log
