cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gemm_a4.py -x -q 2>&1 | tail -3
PROBE_M=139808 timeout 300 python tools/gemm_probe.py 32768,160 fc2,proj,fc2_st,proj_st,qkv,fc1_noact 3 2>&1 | grep -v "amdgpu.ids"
