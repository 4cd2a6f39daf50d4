"""-m gpu: parity of the HIP path AT THE SHAPES THE BENCHMARK RUNS (the bench kernels were only checked at toy widths).

(1) real_b1 fixture (made by the reference at the real widths, one block per stack): strip im2col at 224 / 14, frame attention
    257 x 88, the K = 1408 / 6144 / 2560 / 10240 GEMMs, cross-attention over 2056 keys, the 50272-wide lm_head + argmax.
(2) the four ViT GEMMs at the bench launch shape M = 279 616 (1088 frames x 257 tokens) — `gemm_pp4_kernel` with K = 1408 and
    6144, GELU / residual epilogues — 4096 sampled rows against the oracle's eilev_linear on exactly those rows.
(3) `eilev_vit_forward` on a bench-sized launch (1088 frames) against the same frames run two at a time (the small-tile
    kernels that (1) ties to the reference): the per-frame result must not depend on the launch size.
Tolerances: bf16 storage — as close to the fp32 reference as the reference's own bf16 run (x 1.5 + 1e-3), rel-RMS <= 1e-2;
greedy ids exact.  Every measured distance is recorded (gpurun_out/parity_r02.json -> profiles/).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from eilev_amd import abi
from eilev_amd.configs import blip2_config
from eilev_amd.synth import det_uniform_int, synth_pixels
from oracle import runner as orc

from hip_utils import P, host, models, record_parity, rel_rms, stream_ptr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def real(golden_dir):
    g = np.load(os.path.join(golden_dir, "real_b1.npz"))
    meta = json.loads(str(g["meta"]))
    cfg, _, eng = models(meta["config"])
    px = synth_pixels(sum(sum(c) for c, _ in meta["rows"]), meta["frames"], cfg.vision_config.image_size)
    return g, meta, eng, torch.from_numpy(px).cuda()


def _close_as_reference_bf16(got, fp32, bf16, what):
    ref_dev = float(np.abs(bf16 - fp32).max())
    err = float(np.abs(got - fp32).max())
    rr = rel_rms(got, fp32)
    record_parity("real_b1", **{f"{what}_hip_vs_fp32_maxabs": err, f"{what}_refbf16_vs_fp32_maxabs": ref_dev,
                                f"{what}_hip_vs_fp32_relrms": rr, f"{what}_refbf16_vs_fp32_relrms": rel_rms(bf16, fp32),
                                f"{what}_hip_vs_refbf16_maxabs": float(np.abs(got - bf16).max())})
    assert err <= 1.5 * ref_dev + 1e-3, (what, err, ref_dev)
    assert rr <= 1e-2, (what, rr)


def test_real_width_vit_qformer(real):
    g, meta, eng, px = real
    img, pool = eng.vit(px, want_pooler=True)
    torch.cuda.synchronize()
    rows = host(img).reshape(-1, img.shape[-1])[g["vit_rows"]]
    _close_as_reference_bf16(rows, g["fp32_vit_rows"], g["bf16_vit_rows"], "vit")
    _close_as_reference_bf16(host(pool), g["fp32_pooler"], g["bf16_pooler"], "pooler")
    _close_as_reference_bf16(host(eng.qformer(img)), g["fp32_qformer"], g["bf16_qformer"], "qformer")


def test_real_width_logits_and_greedy_ids(real):
    g, meta, eng, px = real
    ids, vm, am = (torch.from_numpy(g[k]).cuda() for k in ("input_ids", "video_input_mask", "attention_mask"))
    emb = eng.embed_scatter(ids, vm, eng.encode_clips(px))
    _, logits, _ = eng.prefill(emb, am, all_logits=True)
    lg = host(logits)
    _close_as_reference_bf16(lg[:, :, g["logit_cols"]], g["fp32_logits_cols"], g["bf16_logits_cols"], "logits_cols")
    _close_as_reference_bf16(lg[:, -1], g["fp32_logits_last"], g["bf16_logits_last"], "logits_last")
    for graph in (False, True):
        out = eng.greedy_decode(emb, am, meta["new_tokens"], eos_id=-1, use_graph=graph)
        assert np.array_equal(out.cpu().numpy(), g["fp32_greedy_free"]), (graph, out)
    assert np.array_equal(g["bf16_greedy_free"], g["fp32_greedy_free"])


BENCH_M = 1088 * 257
BENCH_GEMMS = [  # name, N, K, epilogue (1 = erf-GELU), residual  — the four GEMMs of a ViT-g block (hf modeling_blip_2.py:328-368)
    ("fc1", 6144, 1408, 1, False), ("fc2", 1408, 6144, 0, True), ("qkv", 4224, 1408, 0, False), ("proj", 1408, 1408, 0, True)]


@pytest.mark.parametrize("name,n,k,epi,resid", BENCH_GEMMS)
def test_bench_shape_gemm_rows_vs_oracle(name, n, k, epi, resid):
    hip = abi.load_hip()
    m = BENCH_M
    g = torch.Generator(device="cuda")
    g.manual_seed({"fc1": 11, "fc2": 12, "qkv": 13, "proj": 14}[name])
    a = torch.randn((m, k), device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn((n, k), device="cuda", generator=g) / k ** 0.5).to(torch.bfloat16)
    b = (0.5 * torch.randn(n, device="cuda", generator=g)).to(torch.bfloat16)
    r = torch.randn((m, n), device="cuda", generator=g).to(torch.bfloat16) if resid else None
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    assert hip.eilev_linear(P(a), P(w), P(b), P(r), P(out), m, n, k, epi, 0, stream_ptr()) == 0
    torch.cuda.synchronize()
    # 4096 rows: the first / last tile rows, the M tail (279 616 = 1092 * 256 + 64) and random rows in between
    rows = np.unique(np.concatenate([np.arange(0, 300), np.arange(m - 300, m), det_uniform_int(f"rows_{name}", (3496,), 0, m)]))
    idx = torch.from_numpy(rows).cuda()
    sa, sr = a[idx].float().cpu().numpy(), (r[idx].float().cpu().numpy() if resid else None)
    wn, bn = w.float().cpu().numpy(), b.float().cpu().numpy()
    ref = np.empty((len(rows), n), np.float32)
    pp = lambda x: None if x is None else np.ascontiguousarray(x).ctypes.data_as(C.c_void_p)
    sa, wn, bn = np.ascontiguousarray(sa), np.ascontiguousarray(wn), np.ascontiguousarray(bn)
    sr = None if sr is None else np.ascontiguousarray(sr)
    assert orc.lib().eilev_linear(pp(sa), pp(wn), pp(bn), pp(sr), pp(ref), len(rows), n, k, epi, 0, None) == 0
    got = out[idx].float().cpu().numpy()
    err = np.abs(got - ref)
    scale = float(np.abs(ref).max())
    record_parity("bench_shape_gemm", **{f"{name}_maxabs": float(err.max()), f"{name}_ref_max": scale, f"{name}_relrms": rel_rms(got, ref),
                                          f"{name}_rows": len(rows)})
    # bf16 output rounding is 2^-9 relative per element; fp32 accumulation over K <= 6144 adds ~1e-6 relative
    assert float((err - 2.0 ** -8 * np.abs(ref)).max()) <= 2e-3 * scale, (name, float(err.max()), scale)
    assert rel_rms(got, ref) <= 3e-3


def test_vit_bench_launch_equals_small_launches():
    cfg, _, eng = models("real_1l")
    frames = 1088
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    px = torch.randn((frames // 8, 3, 8, 224, 224), device="cuda", generator=g).clamp_(-2.5, 2.5).to(torch.bfloat16)
    big = eng.vit(px)                                   # one launch of 1088 frames: the persistent ping-pong GEMMs of the bench
    clips = [0, 1, 67, 134, 135]
    worst = 0.0
    for c in clips:
        small = eng.vit(px[c:c + 1, :, :2])             # 2 frames: the small-tile kernels the real_b1 fixture pins to the reference
        a, b = host(big[c, : 2 * 257]), host(small[0])
        worst = max(worst, rel_rms(a, b))
        assert np.abs(a - b).max() <= 2.0 ** -6 * max(1.0, float(np.abs(b).max())), c
    record_parity("vit_bench_launch", relrms_vs_small_launch=worst)
    assert worst <= 4e-3


def test_prefill_bench_launch_equals_small_launches():
    """The OPT prefill at the bench's launch shape (32 samples x 960 tokens = 30 720 rows: every linear on the persistent 16 x 16 MFMA kernel,
    the fused q|k|v with its q pre-scaling in the lean epilogue, relu, residuals) against the same rows prefilled alone (960 rows: the
    one-wave-per-SIMD kernel with the general epilogue), one block at the real widths; rows 1 and 17 are left-padded.  Both are bf16 paths
    of the same arithmetic: logits agree to bf16 rounding, the argmax exactly where the margin allows."""
    cfg, _, eng = models("real_1l")
    B, L, D = 32, 960, cfg.text_config.hidden_size
    g = torch.Generator(device="cuda")
    g.manual_seed(21)
    emb = (0.5 * torch.randn((B, L, D), device="cuda", generator=g)).to(torch.bfloat16)
    am = torch.ones((B, L), dtype=torch.int64, device="cuda")
    am[1, :300] = 0
    am[17, :7] = 0
    big, _, _ = eng.prefill(emb, am)
    worst = 0.0
    for b in (0, 1, 17, 31):
        small, _, _ = eng.prefill(emb[b:b + 1], am[b:b + 1])
        x, y = host(big[b:b + 1]), host(small)
        worst = max(worst, rel_rms(x, y))
        assert np.abs(x - y).max() <= 2.0 ** -6 * max(1.0, float(np.abs(y).max())), b
    record_parity("prefill_bench_launch", relrms_vs_single_sample=worst)
    assert worst <= 4e-3


def test_real_width_t5_path(golden_dir):
    """BASELINE configs[3] at its real widths (flan-t5-xl, one block per stack): encoder output and logits vs the reference fixture,
    judged like the OPT path; greedy ids equal to the reference's fp32 or bf16 run."""
    g = np.load(os.path.join(golden_dir, "real_t5_b1.npz"))
    meta = json.loads(str(g["meta"]))
    cfg, _, eng = models(meta["config"])
    px = torch.from_numpy(synth_pixels(sum(sum(c) for c, _ in meta["rows"]), meta["frames"], cfg.vision_config.image_size)).cuda()
    t = lambda k: torch.from_numpy(g[k]).cuda()
    emb = eng.embed_scatter(t("input_ids"), t("video_input_mask"), eng.encode_clips(px))
    logits, enc = eng.t5_forward(emb, t("attention_mask"), t("decoder_input_ids"))
    e = host(enc)[:, g["enc_rows"]]
    ref_dev = float(np.abs(g["bf16_enc_rows"] - g["fp32_enc_rows"]).max())
    record_parity("real_t5_b1", enc_hip_vs_fp32_maxabs=float(np.abs(e - g["fp32_enc_rows"]).max()), enc_refbf16_vs_fp32_maxabs=ref_dev,
                  enc_hip_vs_fp32_relrms=rel_rms(e, g["fp32_enc_rows"]), enc_refbf16_vs_fp32_relrms=rel_rms(g["bf16_enc_rows"], g["fp32_enc_rows"]))
    assert np.abs(e - g["fp32_enc_rows"]).max() <= 1.5 * ref_dev + 1e-3
    assert rel_rms(e, g["fp32_enc_rows"]) <= 1.2 * rel_rms(g["bf16_enc_rows"], g["fp32_enc_rows"]) + 2e-3
    lg = host(logits)[:, :, g["logit_cols"]]
    record_parity("real_t5_b1", logits_hip_vs_fp32_relrms=rel_rms(lg, g["fp32_logits_cols"]),
                  logits_refbf16_vs_fp32_relrms=rel_rms(g["bf16_logits_cols"], g["fp32_logits_cols"]))
    assert np.abs(lg - g["fp32_logits_cols"]).max() <= 2.0 * np.abs(g["bf16_logits_cols"] - g["fp32_logits_cols"]).max() + 1e-3
    assert rel_rms(lg, g["fp32_logits_cols"]) <= 1.2 * rel_rms(g["bf16_logits_cols"], g["fp32_logits_cols"]) + 2e-3
    ids = eng.t5_greedy(emb, t("attention_mask"), meta["new_tokens"], eos_id=-1).cpu().numpy()
    assert any(np.array_equal(ids, g[k]) for k in ("fp32_greedy_free", "bf16_greedy_free")), (ids, g["fp32_greedy_free"], g["bf16_greedy_free"])


@pytest.mark.parametrize("B,L", [(2, 24), (3, 24), (5, 24), (8, 24), (20, 24), (32, 24), (16, 530)])
def test_batch_decode_step_at_real_widths_vs_oracle(B, L):
    """B <= 8: the row-dot block of round 4 (gemvm_kernel: rows staged in LDS behind the weight ring, DPP lane sums; attn_decode1_kernel;
    fc2 on gemv_rows_kernel or, at 8 rows, the MFMA kernel).  The batch-17..32 decode block at OPT-2.7B widths (round 4: gemm_skinny5_kernel — q|k|v and fc1 without a K split, out_proj and fc2
    with 2 / 4 K splits through the partial buffer and reduce_ln_kernel; K = 2560 / 10240) against the fp32 ORACLE: after a prefill of L
    positions, one decode step on token t must give the logits a prefill over the L + 1 positions gives for its last row (left padding in
    two rows; ragged M = 20 exercises the row guards).  (16, 530), round 5: 16 rows x 32 heads = 2 workgroups per CU, the form
    attn_decode_loop_kernel takes — 531 keys = two whole 256-key ranges and a ragged one, one row whose first range is padding only."""
    import ctypes as C

    cfg, oracle, eng = models("real_1l")
    d = eng.dims
    rng = np.random.default_rng(5)
    ids = rng.integers(4, 50000, size=(B, L + 1)).astype(np.int64)
    am = np.ones((B, L + 1), np.int64)
    am[1, :5] = 0
    if B > 2:
        am[B - 1, :9] = 0
    if L > 300:
        am[2, :300] = 0
    emb_o = oracle.embed_scatter(ids, None, None)
    ref, _, _ = oracle.prefill(emb_o, am, all_logits=False)
    t = lambda a: torch.from_numpy(a).cuda()
    emb = eng.embed_scatter(t(ids), None, None)
    cap = L + 4
    am_l = t(am[:, :L]).to(torch.int32).contiguous()
    kv = eng.new_kv_cache(B, cap)
    eng.prefill(emb[:, :L].contiguous(), am_l, kv_cache=kv, kv_capacity=cap)
    state = torch.tensor([1, B], dtype=torch.int32, device="cuda")
    tokens = t(ids[:, L]).contiguous()
    finished = torch.zeros(B, dtype=torch.uint8, device="cuda")
    n_valid = am_l.sum(dim=1).to(torch.int32).contiguous()
    out = torch.zeros((B, 4), dtype=torch.int64, device="cuda")
    logits = torch.empty((B, d.vocab), dtype=torch.float32, device="cuda")
    ws = torch.empty(int(eng.lib.eilev_opt_workspace_bytes(C.byref(d), B, 1)), dtype=torch.uint8, device="cuda")
    P = lambda x: C.c_void_p(x.data_ptr())
    rc = eng.lib.eilev_opt_decode_step(C.byref(d), C.byref(eng.pack.opt), P(tokens), P(state), P(am_l), P(n_valid), B, L, P(kv), cap, P(logits), P(finished),
                                       -1, 1, P(out), 4, P(ws), ws.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    got = host(logits)
    for b in range(B):
        assert rel_rms(got[b], ref[b]) <= 1e-2, (b, rel_rms(got[b], ref[b]))
    assert np.array_equal(out[:, 1].cpu().numpy(), got.argmax(-1))


def test_decode_step_at_the_timed_bench_shape_vs_oracle():
    """VERDICT r5 missing 2: the decode step the bench TIMES — 32 rows x ~975 keys at OPT-2.7B widths: attn_decode_loop_kernel<10, 11, 256> over
    four 256-key ranges (the last one ragged), gemm_rows32_kernel with the weights in the STREAM layout and the activations in the row-block
    layout (engine.ensure_stream_layout: what greedy_decode attaches at 17..32 rows), one step eager and the same step replayed from a
    hipGraph.  Two rows are left-padded (one of them by more than a whole key range).  Against the fp32 ORACLE (hf OPTDecoderLayer
    modeling_opt.py:321-396 restated): after a prefill of L positions, a decode step on token t gives the logits a prefill over L + 1
    positions gives for its last row; ids = argmax of the step's own logits; graph replay == eager, bit for bit (logits, ids, cache).
    ref: hf `_sample` (generation/utils.py:2783-2941) via ref:eilev/model/v2.py:318-322."""
    import ctypes as C

    cfg, oracle, eng = models("real_1l")
    d = eng.dims
    B, L = 32, 975
    rng = np.random.default_rng(17)
    ids = rng.integers(4, 50000, size=(B, L + 1)).astype(np.int64)
    am = np.ones((B, L + 1), np.int64)
    am[1, :5] = 0
    am[B - 1, :300] = 0  # its first 256-key range is padding only
    emb_o = oracle.embed_scatter(ids, None, None)
    ref, _, _ = oracle.prefill(emb_o, am, all_logits=False)
    t = lambda a: torch.from_numpy(a).cuda()
    emb = eng.embed_scatter(t(ids), None, None)
    cap = 992  # the bench's capacity: 960 prompt positions + 32 new tokens
    am_l = t(am[:, :L]).to(torch.int32).contiguous()
    kv0 = eng.new_kv_cache(B, cap)
    eng.prefill(emb[:, :L].contiguous(), am_l, kv_cache=kv0, kv_capacity=cap)
    assert eng.ensure_stream_layout(B)
    assert eng.pack.opt.layers_stream and eng.pack.opt.lm_head_stream
    n_valid = am_l.sum(dim=1).to(torch.int32).contiguous()
    ws = torch.empty(int(eng.lib.eilev_opt_workspace_bytes(C.byref(d), B, 1)), dtype=torch.uint8, device="cuda")
    Pp = lambda x: C.c_void_p(x.data_ptr())
    tokens = t(ids[:, L]).contiguous()
    # static buffers of the step (the graph replays on exactly these)
    kv = kv0.clone()
    state = torch.tensor([1, B], dtype=torch.int32, device="cuda")
    finished = torch.zeros(B, dtype=torch.uint8, device="cuda")
    out = torch.zeros((B, 4), dtype=torch.int64, device="cuda")
    logits = torch.empty((B, d.vocab), dtype=torch.float32, device="cuda")

    def launch():
        rc = eng.lib.eilev_opt_decode_step(C.byref(d), C.byref(eng.pack.opt), Pp(tokens), Pp(state), Pp(am_l), Pp(n_valid), B, L, Pp(kv), cap, Pp(logits),
                                           Pp(finished), -1, 1, Pp(out), 4, Pp(ws), ws.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0

    def reset():
        kv.copy_(kv0)
        state.copy_(torch.tensor([1, B], dtype=torch.int32, device="cuda"))
        finished.zero_()
        out.zero_()
        logits.zero_()
        ws.fill_(0x7f)

    reset()
    launch()
    torch.cuda.synchronize()
    eager = (host(logits), out[:, 1].cpu().numpy().copy(), kv.clone())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # (warm-up on the capture stream, as torch asks)
        reset()
        launch()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    reset()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        launch()
    reset()
    graph.replay()
    torch.cuda.synchronize()
    got = host(logits)
    assert np.isfinite(got).all()
    assert np.array_equal(got, eager[0]), "hipGraph replay != eager step"
    assert np.array_equal(out[:, 1].cpu().numpy(), eager[1])
    assert torch.equal(kv, eager[2])
    worst = 0.0
    for b in range(B):
        r = rel_rms(got[b], ref[b])
        worst = max(worst, r)
        assert r <= 1e-2, (b, r)
    assert np.array_equal(eager[1], got.argmax(-1))
    # ids against the ORACLE's own argmax wherever its top-2 margin exceeds what bf16 arithmetic can move (2 x the worst row distance x |logit| scale)
    top2 = np.sort(ref, axis=-1)[:, -2:]
    margin = top2[:, 1] - top2[:, 0]
    floor = 4.0 * worst * np.sqrt((ref ** 2).mean(-1))
    decided = margin > floor
    assert decided.sum() >= B // 2, (margin, floor)
    assert np.array_equal(eager[1][decided], ref.argmax(-1)[decided])
    record_parity("decode_step_32x975", worst_row_relrms=float(worst), rows_decided=int(decided.sum()))


def test_vit_head_major_qkv_launch(probes):
    """Round 5: in ViT launches of >= 512 frames the folded q|k|v GEMM of every block but the first scatters its output into per-head blocks
    ([frame][q, k, v][head]: [token][64] then [token][24]; weight rows reordered to match, include/eilev.h ABI 15) and attn_frame3_kernel's HM
    form stages a head's image from the two contiguous runs.  Three blocks at ViT-g widths, 544 frames: the same bits as the row-major launch
    (probe-build switch), and the small-tile 2-frame launches that the real_b1 fixture pins to the reference."""
    import ctypes as C

    cfg, _, eng = models("real_vit_3l")
    d = eng.dims
    frames = 544
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    px = torch.randn((frames // 8, 3, 8, 224, 224), device="cuda", generator=g).clamp_(-2.5, 2.5).to(torch.bfloat16)
    eng.ensure_vit_fold()
    assert eng.pack.vit.layers_fold_hm
    ws = torch.empty(int(probes.eilev_vit_workspace_bytes(C.byref(d), frames // 8, 8)), dtype=torch.uint8, device="cuda")
    outs = {}
    try:
        for hm in (0, 1):
            probes.eilev_debug_vit_head_major(hm)
            out = torch.empty((frames // 8, 8 * 257, d.v_hidden), dtype=torch.bfloat16, device="cuda")
            assert probes.eilev_vit_forward(C.byref(d), C.byref(eng.pack.vit), P(px), 1, frames // 8, 8, P(out), None, P(ws), ws.numel(), stream_ptr()) == 0
            torch.cuda.synchronize()
            outs[hm] = out
    finally:
        probes.eilev_debug_vit_head_major(1)
    assert torch.isfinite(outs[1].float()).all()
    assert torch.equal(outs[0], outs[1])
    big = eng.vit(px)  # the product library: head-major by default
    assert torch.equal(big, outs[1])
    worst = 0.0
    for c in (0, 33, 67):
        small = eng.vit(px[c:c + 1, :, :2])
        a, b = host(big[c, : 2 * 257]), host(small[0])
        worst = max(worst, rel_rms(a, b))
        assert np.abs(a - b).max() <= 2.0 ** -5 * max(1.0, float(np.abs(b).max())), c
    record_parity("vit_head_major_launch", relrms_vs_small_launch=worst)
    assert worst <= 6e-3
