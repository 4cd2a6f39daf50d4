# round 6, call 21: what the driver runs at round end: smoke() and the default bench command
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python bench.py 2> gpurun_out/r06_bench_default.err | tail -1 > gpurun_out/r06_bench_default.json
python -c "
import json; d=json.load(open('gpurun_out/r06_bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d.get('verified'), d['cpu_baseline']['value'], d['cpu_baseline'].get('extrapolated'))"
