cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
# T5 decode: rows32 KS = 8 for the wide matrices only (default) vs also for the N = 2048 ones (r32all) vs HEAD (split-K kernels everywhere)
cp eilev_amd/csrc/libeilev_hip.so /tmp/base.so
for v in base r32all head base r32all; do
  if [ $v = base ]; then cp /tmp/base.so eilev_amd/csrc/libeilev_hip.so; else cp eilev_amd/csrc/libeilev_hip_$v.so eilev_amd/csrc/libeilev_hip.so; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-strong --no-pmc --lm t5xl 2>/dev/null | tail -1 > /tmp/t5_$v.json
  python -c "
import json; d=json.load(open('/tmp/t5_$v.json')); print('$v', d['value'], d['phases_rank0']['prefill_ms_per_step'], 'decode ms/token', d['phases_rank0']['decode_ms_per_token'])"
done > $O/r05_t5_decode_rows32_ab.log 2>&1
cp /tmp/base.so eilev_amd/csrc/libeilev_hip.so
cat $O/r05_t5_decode_rows32_ab.log
