#!/usr/bin/env python
"""GPU probe: time eilev_linear on the ViT/OPT GEMM shapes; variants interleaved, several rounds (min / median).

    [PROBE_M=rows] python tools/gemm_probe.py <flags,flags,...> <shape,shape,...> <rounds>

flags (eilev_debug_gemm_flags): 4 register-staged reference kernel; 8 old skinny kernel; (n << 4) force tile config n
(1: 256x256 per-tile kernel, 2: 256x128 two stages, 3: 256x128 one stage x 2 workgroups/CU, 4: 128x128, 9: persistent ping-pong
kernel pp4, 12: one-wave-per-SIMD kernel w6, 13: 64x128 two-wave tiles, 14: 128x128); 1024 no epilogue; 2048 no store phase; 4096 / 8192 alias all A / W rows onto row 0
(cache-resident operand — also changes the MFMA data statistics and with them the clock: not a memory-system measurement);
131072 alias all output rows onto row 0; 524288 no half-tile path; 1048576 per-tile kernel for N = 1408; 2097152 never pick w6;
16777216 pp4 without the lean epilogue / second pre-staged K-step; 65536 half tiles dealt as single tiles (round 6: the deal before half-tile pairing); (n << 22) tile-group height override (1: 4 rows, 2: 8, 3: 16).
PROBE_M / PROBE_MOPT override the row count of the ViT / OPT shapes.
"""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from eilev_amd import abi

_flags_req = [x for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"]) if int(x)]
if os.environ.get("EILEV_PROBE_ON_PRODUCT_LIB") and not _flags_req:
    pass  # flags 0 only on libeilev_hip.so itself: what bench.py's live traffic measurement profiles
else:
    abi.use_probes()  # the eilev_debug_* switches live in the probe build only (libeilev_hip_probes.so)

lib = abi.load_hip()
raw = C.CDLL(abi.HIP_LIB_PATH)
if not hasattr(raw, "eilev_debug_gemm_flags"):  # product library: no switches (flags 0 only, see above)
    class _NoSwitch:
        product = True  # no register-staged reference kernel to compare with: the self-comparison below is skipped, and says so

        def eilev_debug_gemm_flags(self, f):
            assert f == 0, f
    raw = _NoSwitch()
    print("product library: timing only (the max-error check against the register-staged kernel needs the probe build)")
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
NOBIAS = bool(os.environ.get("PROBE_NOBIAS"))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ALL = {"fc1": (34952, 6144, 1408, 1, False), "fc2": (34952, 1408, 6144, 0, True), "qkv": (34952, 4224, 1408, 0, False),
       "proj": (34952, 1408, 1408, 0, True), "opt_fc1": (7680, 10240, 2560, 2, False), "opt_qkv": (7680, 7680, 2560, 0, False),
       "qf_kv": (34952, 1536, 1408, 0, False), "opt_fc2": (7680, 2560, 10240, 0, True),
       "fc2_k6208": (34952, 1408, 6208, 0, True), "fc2_k6080": (34952, 1408, 6080, 0, True), "fc1_k1472": (34952, 6144, 1472, 0, False),
       "opt_out": (7680, 2560, 2560, 0, True), "qf_dense": (544, 768, 768, 0, True), "qf_fi": (544, 3072, 768, 0, False),
       "fc1_noact": (34952, 6144, 1408, 0, False), "fc1_relu": (34952, 6144, 1408, 2, False),
       # what the folded-LayerNorm ViT blocks run (eilev_linear_lnfold / eilev_linear_stats): consumer (_ln) and statistics producer (_st)
       "fc1_ln": (34952, 6144, 1408, 1, False), "qkv_ln": (34952, 4224, 1408, 0, False), "proj_st": (34952, 1408, 1408, 0, True),
       "fc2_st": (34952, 1408, 6144, 0, True),
       # round 4 diagnostic: the fc2 K loop without a half tile column (5 / 6 / 8 whole column tiles) — is the 5.5-column drift what costs fc2?
       # round 5: the flan-t5-xl encoder linears of a bench step (32 samples x 960 tokens; no biases: PROBE_NOBIAS=1)
       "t5_qkv": (30720, 6144, 2048, 0, False), "t5_o": (30720, 2048, 2048, 0, True), "t5_wi": (30720, 10240, 2048, 0, False),
       "t5_wo": (30720, 2048, 5120, 0, True),
       "fc2_n1280": (34952, 1280, 6144, 0, True), "fc2_n1536": (34952, 1536, 6144, 0, True), "fc2_n2048": (34952, 2048, 6144, 0, True),
       # round 6: what does a half tile cost?  One column of half tiles (N = 128) against one column of whole tiles (N = 256), PROBE_M = 262144 = 4 exact rounds
       "fc2_n128": (34952, 128, 6144, 0, True), "fc2_n256": (34952, 256, 6144, 0, True), "proj_n128": (34952, 128, 1408, 0, True), "proj_n256": (34952, 256, 1408, 0, True)}
flags_list = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"])]
names = sys.argv[2].split(",") if len(sys.argv) > 2 else list(ALL)
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for name in names:
    m, n, k, epi, resid = ALL[name]
    if os.environ.get("PROBE_M") and m == 34952:
        m = int(os.environ["PROBE_M"])
    if os.environ.get("PROBE_MOPT") and m == 7680:
        m = int(os.environ["PROBE_MOPT"])
    if os.environ.get("PROBE_MT5") and name.startswith("t5_"):
        m = int(os.environ["PROBE_MT5"])
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda").to(torch.bfloat16)
    r = torch.randn(m, n, device="cuda").to(torch.bfloat16) if resid else None
    o = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    if name.endswith("_ln"):
        cs = torch.randn(n, device="cuda")
        rows = torch.stack([torch.rand(m, device="cuda") + 0.5, torch.randn(m, device="cuda") * 0.1], 1).contiguous()
        call = lambda dst: lib.eilev_linear_lnfold(P(a), P(w), P(b), P(cs), P(rows), P(dst), m, n, k, epi, st())
    elif name.endswith("_st"):
        stats = torch.empty(((n + 63) // 64, m, 2), device="cuda")
        call = lambda dst: lib.eilev_linear_stats(P(a), P(w), P(b), P(r), P(dst), m, n, k, P(stats), st())
    else:
        call = lambda dst: lib.eilev_linear(P(a), P(w), P(b), P(r), P(dst), m, n, k, epi, 0, st())
    fold = name.endswith(("_ln", "_st")) or getattr(raw, "product", False)  # (no register-staged reference of these: tests/test_ln_fold.py checks them)
    if not fold:
        raw.eilev_debug_gemm_flags(4)   # reference: register-staged kernel
    ref = torch.empty_like(o)
    if not fold:
        call(ref)
    for flags in flags_list:
        if fold:
            break
        if flags & 3:
            continue
        raw.eilev_debug_gemm_flags(flags)
        o.zero_()
        call(o)
        torch.cuda.synchronize()
        d = (o.float() - ref.float()).abs().max().item()
        if d != 0.0:
            print(f"{name}: flags={flags} MISMATCH vs register-staged kernel: max abs diff {d}")
    base = None  # every variant against the first one (the LayerNorm-folding shapes have no register-staged reference)
    for flags in flags_list:
        raw.eilev_debug_gemm_flags(flags)
        o.zero_()
        call(o)
        torch.cuda.synchronize()
        if base is None:
            base = o.float().clone()
        else:
            d = (o.float() - base).abs().max().item()
            print(f"{name}: flags={flags} max abs diff vs flags={flags_list[0]}: {d:.4g} (rms of the output {base.pow(2).mean().sqrt().item():.4g})")
    del base
    times = {f: [] for f in flags_list}
    for rd in range(rounds + 1):
        for flags in flags_list:
            raw.eilev_debug_gemm_flags(flags)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call(o)
            e1.record(); torch.cuda.synchronize()
            if rd:
                times[flags].append(e0.elapsed_time(e1) / 5)
    for flags in flags_list:
        t = times[flags]
        med, mn = statistics.median(t), min(t)
        print(f"{name:8s} M={m} N={n} K={k} flags={flags:3d} med {med*1e3:7.1f} us {2*m*n*k/med/1e9:7.1f} TF/s | min {mn*1e3:7.1f} us {2*m*n*k/mn/1e9:7.1f} TF/s", flush=True)
raw.eilev_debug_gemm_flags(0)
