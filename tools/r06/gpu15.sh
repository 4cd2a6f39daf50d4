# round 6, call 15: GELU as x * Phi(x) (7 issue-equivalents) against the degree-8 bump form (8.9): same-box A/B of the two builds, accuracy tests
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
C=$R/eilev_amd/csrc
AB_SHAPES=fc1_ln,fc1,fc1_noact timeout 900 python $R/tools/gemm_ab.py $C/libeilev_hip_gelu8.so $C/libeilev_hip.so 279616 7 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" > $O/r06_gelu_phi_ab.log
cat $O/r06_gelu_phi_ab.log
cd $R && timeout 900 python -m pytest tests/test_hip_real_shapes.py tests/test_ln_fold.py tests/test_hip_kernels.py -q -m gpu -x -k "bench_shape_gemm or ln_fold or folded or gemm or linear" 2>&1 | tail -3
