cd $GRAFT_REPO_ROOT
for v in ${A4_VARS:-0 1 2 3}; do
  echo "=== A4 variant $v"
  EILEV_A4_VAR=$v PROBE_M=279616 timeout 300 python tools/gemm_probe.py ${A4_FLAGS:-0,160,1184} ${A4_SHAPES:-fc1_noact,fc2,qkv} 3 2>&1 | grep -v "amdgpu.ids\|flags=1184 MISMATCH\|flags=1024 MISMATCH"
done
