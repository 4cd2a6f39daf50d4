cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["vit_gemm_ms_per_step"], json.dumps(d["roofline"]["vit_gemm_us_and_tflops"]))'
for r in 1 2; do
echo "== new"; timeout 400 python bench.py --no-cpu-baseline --no-strong --no-verify --no-pmc 2>/dev/null | tail -1 | python -c "$P"
cp eilev_amd/csrc/libeilev_hip.so /tmp/new.so; cp tools/probes/libeilev_prev.so eilev_amd/csrc/libeilev_hip.so
echo "== prev"; timeout 400 python bench.py --no-cpu-baseline --no-strong --no-verify --no-pmc 2>/dev/null | tail -1 | python -c "$P"
cp /tmp/new.so eilev_amd/csrc/libeilev_hip.so
done
