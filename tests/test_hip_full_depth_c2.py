"""-m gpu: the HEADLINE shape at FULL DEPTH against the reference — one 16-shot sample (17 clips x 8 frames, L = 960: the bench's token
layout) plus a second, shorter row (3 clips, L = 128) LEFT-padded to 960, through all 39 ViT-g + 12 Q-Former + 32 OPT-2.7B blocks at the
real widths, against the reference's own fp32 and bf16 runs (tests/golden/full_c2.npz, produced by `tools/make_goldens.py full_c2` from
/root/reference in the build container, ~25 minutes of CPU; weights by recipe: eilev_amd.synth 'varied', seed in the meta).

Until round 5 the C2 shape (batch > 1, left padding, 20 clips in one launch, L = 960) was checked at depth only HIP-vs-own-oracle inside
bench.py (VERDICT r4 missing 2); full_c1 pins the 1-clip / L = 48 shape.  Checked: the full-vocabulary last-row prefill logits of both rows,
the 2 x 32 greedy ids — the reference's fp32 and bf16 runs emit the same 64 ids, but the tightest of their 64 top-2 margins is of the
size of the bf16 deviation, so ids are compared up to the first near-tie of the REFERENCE (oracle/parity.py) — the eight leading logits of
every step before that, and a FORCED continuation of 24 pseudo-random tokens on both rows (batch-2 decode step with left padding).
Tolerance as everywhere: HIP-vs-fp32 error <= 1.5 x the reference's own bf16-vs-fp32 error (+ slack)."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from eilev_amd.configs import blip2_config
from eilev_amd.statedict import state_dict_shapes
from eilev_amd.synth import synth_param_torch, synth_pixels
from hip_utils import host, record_parity, rel_rms
from oracle.parity import greedy_ids_vs_reference

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(golden_dir):
    from eilev_amd.engine import HipEngine

    g = np.load(os.path.join(golden_dir, "full_c2.npz"))
    meta = json.loads(str(g["meta"]))
    cfg = blip2_config(meta["config"])
    sd = {}
    for k, shp in state_dict_shapes(cfg).items():  # the recipe evaluated ON the device (bit-identical to numpy: tests/test_synth_torch.py)
        sd[k] = synth_param_torch(k, shp, meta["weight_mode"], meta["weight_seed"], device="cuda").to(torch.bfloat16)
    eng = HipEngine(cfg, sd, device="cuda")
    del sd
    nclips = sum(sum(c) for c, _ in meta["rows"])
    assert nclips == 20 and g["input_ids"].shape == (2, 960)
    px = synth_pixels(nclips, meta["frames"], cfg.vision_config.image_size)
    feats = eng.encode_clips(torch.from_numpy(px).cuda())
    emb = eng.embed_scatter(torch.from_numpy(g["input_ids"]).cuda(), torch.from_numpy(g["video_input_mask"]).cuda(), feats)
    am = torch.from_numpy(g["attention_mask"]).cuda()
    return g, meta, eng, emb, am


def test_full_depth_c2_prefill_logits(full):
    g, meta, eng, emb, am = full
    last, _, _ = eng.prefill(emb, am)
    got, r32, r16 = host(last), g["fp32_logits_last"], g["bf16_logits_last"]
    scale = float(np.abs(r32).max())
    err = float(np.abs(got - r32).max())
    record_parity("full_depth[c2]", prefill_last_hip_vs_fp32_rel_rms=rel_rms(got, r32), prefill_last_refbf16_vs_fp32_rel_rms=rel_rms(r16, r32),
                  prefill_last_hip_vs_refbf16_rel_rms=rel_rms(got, r16), prefill_last_hip_vs_fp32_maxabs=err,
                  prefill_last_refbf16_vs_fp32_maxabs=float(np.abs(r16 - r32).max()), logit_std=float(r32.std()))
    assert err <= 1.5 * float(np.abs(r16 - r32).max()) + 2e-3 * scale, (err, float(np.abs(r16 - r32).max()))
    for b in range(2):  # each row on its own: the long one and the left-padded short one
        assert rel_rms(got[b], r32[b]) <= 1.5 * rel_rms(r16[b], r32[b]) + 1e-3, b
    assert np.array_equal(got.argmax(-1), r32.argmax(-1))


@pytest.mark.parametrize("use_graph", [True, False])
def test_full_depth_c2_greedy_ids_and_step_logits(full, use_graph):
    g, meta, eng, emb, am = full
    n = meta["new_tokens"]
    assert np.array_equal(g["fp32_greedy_free"], g["bf16_greedy_free"])
    if use_graph:
        ids = eng.greedy_decode(emb, am, n, eos_id=-1, use_graph=True).cpu().numpy()
        verdict = greedy_ids_vs_reference(ids, g)
        record_parity("full_depth[c2]", greedy_graph=verdict)
        assert verdict["ok"], verdict
        return
    ids, steps = eng.greedy_decode(emb, am, n, eos_id=-1, use_graph=False, return_step_logits=True)
    ids = ids.cpu().numpy()
    verdict = greedy_ids_vs_reference(ids, g)
    assert verdict["ok"], verdict
    first_flip = {f["row"]: f["step"] for f in verdict["flips"]}
    worst = 0.0
    for k in range(n):
        lg = host(steps[k])
        for b in range(ids.shape[0]):
            if k > first_flip.get(b, n):  # past a legitimate flip the row is on another trajectory
                continue
            top = g["fp32_step_logits_top8_ids"][k, b]
            mine, r32 = lg[b, top], g["fp32_step_logits_top8"][k, b]
            same = g["bf16_step_logits_top8_ids"][k, b] == top
            ref_dev = float(np.abs(g["bf16_step_logits_top8"][k, b] - r32)[same].max()) if same.any() else 0.0
            err = float(np.abs(mine - r32).max())
            worst = max(worst, err)
            assert err <= 1.5 * ref_dev + 2e-3 * float(np.abs(r32).max()) + 0.05, (k, b, err, ref_dev)
    record_parity("full_depth[c2]", greedy_step_top8_max_abs_err=worst, greedy_eager=verdict)


def test_full_depth_c2_forced_continuation(full):
    """24 given tokens per row fed one by one through eilev_opt_decode_step (batch 2, the second row left-padded) after a prefill of the
    prompts: the logits of every step against one forward of the reference over prompt + tokens."""
    g, meta, eng, emb, am = full
    d = eng.dims
    forced = g["forced_tokens"]
    B, L = g["input_ids"].shape
    n = forced.shape[1]
    cap = L + n + 1
    kv = eng.new_kv_cache(B, cap)
    last, _, _ = eng.prefill(emb, am, kv_cache=kv, kv_capacity=cap)
    rows = [host(last)]
    am32 = am.to(torch.int32).contiguous()
    n_valid = am32.sum(dim=1).to(torch.int32).contiguous()
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    tokens = torch.zeros(B, dtype=torch.int64, device="cuda")
    finished = torch.zeros(B, dtype=torch.uint8, device="cuda")
    out = torch.zeros((B, n + 2), dtype=torch.int64, device="cuda")
    logits = torch.empty((B, d.vocab), dtype=torch.float32, device="cuda")
    ws = torch.empty(int(eng.lib.eilev_opt_workspace_bytes(C.byref(d), B, 1)), dtype=torch.uint8, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    for j in range(n):
        tokens.copy_(torch.from_numpy(forced[:, j]).cuda())
        state.copy_(torch.tensor([j + 1, B], dtype=torch.int32))
        finished.zero_()
        rc = eng.lib.eilev_opt_decode_step(C.byref(d), C.byref(eng.pack.opt), P(tokens), P(state), P(am32), P(n_valid), B, L, P(kv), cap, P(logits), P(finished),
                                           -1, 1, P(out), n + 2, P(ws), ws.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        rows.append(host(logits))
    worst = 0.0
    for j in range(n + 1):
        top = g["fp32_forced_top8_ids"][:, j]
        r32, mine = g["fp32_forced_top8"][:, j], np.take_along_axis(rows[j], top, -1)
        same = g["bf16_forced_top8_ids"][:, j] == top
        ref_dev = float(np.abs(g["bf16_forced_top8"][:, j] - r32)[same].max()) if same.any() else 0.0
        err = float(np.abs(mine - r32).max())
        worst = max(worst, err)
        assert err <= 1.5 * ref_dev + 2e-3 * float(np.abs(r32).max()) + 0.05, (j, err, ref_dev)
    r32, r16 = g["fp32_forced_logits_row12"], g["bf16_forced_logits_row12"]
    for b in range(B):
        assert rel_rms(rows[12][b], r32[b]) <= 1.5 * rel_rms(r16[b], r32[b]) + 1e-3, b
    record_parity("full_depth[c2]", forced_top8_max_abs_err=worst, forced_row12_hip_vs_fp32_rel_rms=rel_rms(rows[12], r32),
                  forced_row12_refbf16_vs_fp32_rel_rms=rel_rms(r16, r32))
