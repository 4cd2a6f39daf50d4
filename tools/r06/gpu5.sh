# round 6, call 5: the kernel menu at the batch-1 prefill shapes (M = 960): default dispatch, pp4 (144), w6 (192), 256x128 two stages (32), 128x128 (64), 64x128 (208)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
PROBE_MOPT=960 timeout 900 python $R/tools/gemm_probe.py 0,144,192,32,64,208 opt_qkv,opt_out,opt_fc1,opt_fc2 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids\|MISMATCH" > $O/r06_m960_menu.log
cat $O/r06_m960_menu.log
