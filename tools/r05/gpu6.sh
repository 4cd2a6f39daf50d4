cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
# T5 encoder linears: default dispatch (0) vs forced persistent ping-pong (9 << 4) vs forced one-wave-per-SIMD w6 (12 << 4)
PROBE_NOBIAS=1 timeout 600 python $R/tools/gemm_probe.py 0,144,192 t5_qkv,t5_o,t5_wi,t5_wo 5 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" > $O/r05_t5_gemm_probe.log
cat $O/r05_t5_gemm_probe.log
