# round 6, call 27: K-lock-step probe, sync period 8 / 16 / 32 / 96 K-steps (flags 262144 + 65536 / 33554432), on the product's instances
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
PROBE_M=279616 timeout 900 python $R/tools/gemm_probe.py 0,262144,327680,33816576,33882112 fc2_st,proj_st,fc1_ln 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids\|max abs" > $O/r06_xcd_lockstep_period.log
cat $O/r06_xcd_lockstep_period.log
