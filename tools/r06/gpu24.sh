# round 6, call 24: is the 1-ulp difference of qkv_ln under the lock-step probe a run-to-run difference of the kernel itself?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
{
for f in 0,0 262144,262144 0,262144 0,0; do
PROBE_M=279616 timeout 300 python $R/tools/gemm_probe.py $f qkv_ln 1 2>&1 | grep "max abs diff"
done
} > $O/r06_qkv_determinism.log 2>&1
cat $O/r06_qkv_determinism.log
