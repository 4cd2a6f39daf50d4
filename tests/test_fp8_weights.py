"""fp8 (e4m3) weight-only linear: format (oracle encode / decode vs torch's float8_e4m3fn cast), quantiser, and on the GPU the
HIP kernels (byte-streaming skinny kernel for M <= 32, expand + bf16 kernels above) against the oracle on the same bytes."""
import ctypes as C

import numpy as np
import pytest
import torch

from eilev_amd import quant
from oracle import runner as orc


def _olib():
    L = orc.lib()
    L.eilev_e4m3_decode.restype = C.c_float
    L.eilev_e4m3_decode.argtypes = [C.c_uint8]
    L.eilev_e4m3_encode.restype = C.c_uint8
    L.eilev_e4m3_encode.argtypes = [C.c_float]
    return L


def test_e4m3_decode_matches_torch_for_all_256_bytes():
    L = _olib()
    b = torch.arange(256, dtype=torch.uint8)
    ref = b.view(torch.float8_e4m3fn).to(torch.float32).numpy()
    got = np.array([L.eilev_e4m3_decode(int(i)) for i in range(256)], np.float32)
    nan = np.isnan(ref)
    assert np.array_equal(nan, np.isnan(got)) and np.array_equal(ref[~nan], got[~nan])


def test_e4m3_encode_matches_torch_cast():
    L = _olib()
    g = torch.Generator().manual_seed(1)
    x = torch.cat([torch.randn(20000, generator=g) * 100, torch.randn(20000, generator=g), torch.randn(20000, generator=g) * 0.01,
                   torch.tensor([0.0, -0.0, 448.0, -448.0, 447.9, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -10, 3 * 2.0 ** -10, 0.0625, 0.017578125])])
    x = x.clamp(-448, 448)
    ref = x.to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = np.array([L.eilev_e4m3_encode(float(v)) for v in x.numpy()], np.uint8)
    # +0 / -0 of tiny inputs: compare decoded values and signs of non-zeros
    dec = lambda a: torch.from_numpy(a.copy()).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    assert np.array_equal(dec(ref), dec(got))


def test_quantiser_round_trip_error_and_scale():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(64, 256, generator=g) * torch.rand(64, 1, generator=g) * 3
    w[5] = 0
    q, s = quant.quantize_e4m3_per_channel(w)
    assert q.dtype == torch.uint8 and s.dtype == torch.float32 and s[5] == 1
    dq = quant.dequantize(q, s)
    # e4m3 has 3 mantissa bits: relative error <= 2^-4 for normal values, absolute <= scale * 2^-10 near zero
    assert torch.all((dq - w).abs() <= w.abs() * 2.0 ** -4 + s[:, None] * 2.0 ** -10 + 1e-12)
    assert torch.allclose(dq.abs().amax(1)[s != 1], w.abs().amax(1)[s != 1])  # the row maximum is represented exactly (448 * scale)


def _case(m, n, k, epi, bias, resid, out_f32=False, seed=0):
    from eilev_amd import abi

    g = torch.Generator().manual_seed(seed)
    a = torch.randn(m, k, generator=g).to(torch.bfloat16)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    q, s = quant.quantize_e4m3_per_channel(w)
    b = (0.5 * torch.randn(n, generator=g)).to(torch.bfloat16) if bias else None
    r = torch.randn(m, n, generator=g).to(torch.bfloat16) if resid else None
    f = lambda t: None if t is None else np.ascontiguousarray(t.to(torch.float32).numpy())
    ref = np.empty((m, n), np.float32)
    pp = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    af, bf_, rf, qn, sn = f(a), f(b), f(r), q.numpy(), s.numpy()
    assert orc.lib().eilev_linear_w8(pp(af), pp(qn), pp(sn), pp(bf_), pp(rf), pp(ref), m, n, k, epi, 0, None, 0, None) == 0
    d = lambda t: None if t is None else t.cuda()
    got = quant.linear_w8(d(a), d(q), d(s), d(b), d(r), epilogue=epi, out_dtype=torch.float32 if out_f32 else None, lib=abi.load_hip())
    got = got.to(torch.float32).cpu().numpy()
    err = np.abs(got - ref).max()
    tol = (2e-4 if out_f32 else 1e-2) * np.abs(ref).max()
    assert err <= tol, (m, n, k, epi, err, tol)


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k", [(1, 160, 256), (8, 2560, 2560), (17, 7680, 2560), (32, 2560, 10240), (32, 1000, 512), (3, 50272, 2560)])
def test_linear_w8_decode_shapes(m, n, k):
    _case(m, n, k, 0, bias=True, resid=True)


@pytest.mark.gpu
def test_linear_w8_fp32_logits_and_activations():
    _case(8, 1024, 2560, 0, bias=False, resid=False, out_f32=True)
    _case(32, 2048, 2560, 2, bias=True, resid=False)
    _case(16, 768, 768, 1, bias=True, resid=False)


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k,epi,resid", [(300, 200, 128, 0, True), (1000, 1408, 1408, 1, False), (960, 2560, 2560, 2, False),
                                              (8229, 2048, 320, 0, True), (40, 256, 192, 0, False)])
def test_linear_w8_prefill_shapes_expand_path(m, n, k, epi, resid):
    _case(m, n, k, epi, bias=True, resid=resid)


# ------------------------------------------------------------------------------ fp8 weights inside the OPT pipeline (GPU)
def _fp8_models(cfg_name):
    """HIP engine with fp8 OPT linears + oracle on the DEQUANTISED weights (same bytes: the quantiser is deterministic)."""
    from eilev_amd.configs import blip2_config
    from eilev_amd.engine import HipEngine
    from oracle.runner import OracleModel, synth_state_dict

    cfg = blip2_config(cfg_name)
    sd = synth_state_dict(cfg, "fanin")
    eng = HipEngine(cfg, {k: torch.from_numpy(v).cuda() for k, v in sd.items()}, device="cuda", lm_weights="fp8")
    sdq = dict(sd)
    for k, v in sd.items():
        if k.startswith("language_model.model.decoder.layers.") and k.endswith(("q_proj.weight", "k_proj.weight", "v_proj.weight",
                                                                                 "out_proj.weight", "fc1.weight", "fc2.weight")):
            q, s = quant.quantize_e4m3_per_channel(torch.from_numpy(v))
            sdq[k] = quant.dequantize(q, s).numpy()
    return cfg, OracleModel(cfg, sdq, emulate_bf16=True), eng


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_b2", "mid_b2"])
def test_opt_pipeline_with_fp8_weights_matches_oracle_on_dequantised_weights(golden_dir, name):
    """Inputs of the golden cases (the goldens' outputs do not apply: other weights).  Prefill runs the expand path, the decode
    steps the byte-streaming kernel; both against the oracle on the dequantised weights, greedy ids exact."""
    from hip_utils import host, load_case, rel_rms

    g, meta, px = load_case(golden_dir, name)
    cfg, oracle, eng = _fp8_models(meta["config"])
    feats = eng.encode_clips(torch.from_numpy(px).cuda())
    emb = eng.embed_scatter(torch.from_numpy(g["input_ids"]).cuda(), torch.from_numpy(g["video_input_mask"]).cuda(), feats)
    am = torch.from_numpy(g["attention_mask"]).cuda()
    _, alll, _ = eng.prefill(emb, am.to(torch.int32), all_logits=True, last_logits=False)
    ref = oracle.forward_logits(px, g["input_ids"], g["attention_mask"], g["video_input_mask"])
    valid = g["attention_mask"] == 1
    assert rel_rms(host(alll)[valid], ref[valid]) <= 2e-2
    ids, steps = eng.greedy_decode(emb, am, meta["new_tokens"], eos_id=-1, use_graph=False, return_step_logits=True)
    oids, osteps = oracle.generate(px, g["input_ids"], g["attention_mask"], g["video_input_mask"], meta["new_tokens"], eos_id=-1,
                                   return_logits=True)
    hid = ids.cpu().numpy()
    alive = np.ones(hid.shape[0], bool)  # rows whose token history still equals the oracle's
    for t, (a_, b_) in enumerate(zip(steps, osteps)):
        assert rel_rms(host(a_)[alive], b_[alive]) <= 2e-2
        for r in np.nonzero(alive)[0]:
            if hid[r, t] != oids[r, t]:
                # a different arg-max is only acceptable at a near tie of the oracle's own logits (random tiny model, fp8 noise)
                assert b_[r].max() - b_[r, hid[r, t]] <= 3e-2 * b_[r].std(), (r, t, hid[r], oids[r])
                alive[r] = False
    assert alive.sum() >= 1
    ids_g = eng.greedy_decode(emb, am, meta["new_tokens"], eos_id=-1, use_graph=True)  # and under the captured graph
    assert torch.equal(ids_g, ids)


@pytest.mark.gpu
def test_model_class_switches_to_fp8_weights(golden_dir):
    from eilev_amd.configs import blip2_config
    from eilev_amd.model.v2 import VideoBlipForConditionalGeneration
    from hip_utils import load_case

    g, meta, px = load_case(golden_dir, "tiny_b2")
    torch.manual_seed(0)
    model = VideoBlipForConditionalGeneration(blip2_config(meta["config"])).to(torch.bfloat16).cuda().eval()
    kw = dict(input_ids=torch.from_numpy(g["input_ids"]).cuda(), attention_mask=torch.from_numpy(g["attention_mask"]).cuda(),
              pixel_values=torch.from_numpy(px).cuda().to(torch.bfloat16), video_input_mask=torch.from_numpy(g["video_input_mask"]).cuda())
    ref = model(**kw).logits.float()
    model.hip_lm_weights = "fp8"
    got = model(**kw).logits.float()
    assert model.engine().lm_weights == "fp8"
    rel = ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    assert 0 < rel < 0.2  # e4m3 weights (3 mantissa bits) perturb the logits, they do not scramble them
