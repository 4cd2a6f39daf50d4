# same-box A/B of the bench with the two frame-attention kernels (probe flag 32 = forbid attn_frame3_kernel)
cd $GRAFT_REPO_ROOT
cat > /tmp/run_bench_flag.py <<'PY'
import ctypes as C, sys, json, io, contextlib, os
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-strong", "--no-verify", "--no-pmc"]
flag = int(os.environ.get("ATTN_FLAG", "0"))
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
from eilev_amd import abi
abi.load_hip()
C.CDLL(abi.HIP_LIB_PATH).eilev_debug_attn_v1(flag << 1)
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print(flag, d["value"], d["ms_per_step"], d["phases_rank0"]["encode_ms_per_step"])
PY
for r in 1 2 3; do
  ATTN_FLAG=0 timeout 400 python /tmp/run_bench_flag.py 2>&1 | tail -1
  ATTN_FLAG=32 timeout 400 python /tmp/run_bench_flag.py 2>&1 | tail -1
done
