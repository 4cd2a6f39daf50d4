"""Device-side process(): Pillow-bicubic resize + rescale + normalise of uint8 frames (eilev_process_frames).

not gpu: the host tables and the CPU oracle against (1) the committed golden vectors made with the HF image processor the
reference wraps (tools/make_process_golden.py), (2) PIL / the HF processor themselves when importable.  gpu: the HIP
kernels through the C ABI against the oracle, bit-exact (integer / table work)."""
import ctypes as C
import os

import numpy as np
import pytest

from eilev_amd import preprocess
from oracle import runner as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "process_frames.npz")
CASES = ["down", "up", "wide_only", "identity", "tall_only"]


@pytest.mark.parametrize("n_in,n_out", [(224, 224), (341, 224), (640, 224), (1280, 224), (100, 224), (13, 24), (83, 24), (50, 24)])
def test_coefficient_tables_python_equals_oracle(n_in, n_out):
    L = orc.lib()
    L.eilev_resample_coeffs.restype = C.c_int
    L.eilev_resample_coeffs.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    coef, bounds = preprocess.resample_coeffs(n_in, n_out)
    ks = L.eilev_resample_coeffs(n_in, n_out, None, None)
    assert ks == coef.shape[1]
    c2 = np.zeros_like(coef)
    b2 = np.zeros_like(bounds)
    L.eilev_resample_coeffs(n_in, n_out, c2.ctypes.data, b2.ctypes.data)
    assert np.array_equal(coef, c2) and np.array_equal(bounds, b2)
    # every row's taps sum to 1.0 in 22-bit fixed point up to rounding of the individual taps
    assert np.all(np.abs(coef.sum(1) - (1 << preprocess.PRECISION_BITS)) <= coef.shape[1])


@pytest.mark.parametrize("case", CASES)
def test_oracle_equals_reference_golden(case):
    g = np.load(GOLD)
    got = orc.process_frames(g[f"{case}_video"], size=int(g[f"{case}_size"]))
    ref = g[f"{case}_pixel_values"]
    assert got.shape == ref.shape and got.dtype == np.float32
    assert np.array_equal(got, ref)  # bit-exact


def test_oracle_equals_pil_and_hf_processor_live():
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    lut = preprocess.normalize_lut()
    for (h, w) in [(256, 341), (480, 640), (100, 150), (224, 300)]:
        video = rng.integers(0, 256, (1, 3, 2, h, w), dtype=np.uint8)
        got = orc.process_frames(video, size=224)
        for t in range(2):
            img = PIL.fromarray(np.ascontiguousarray(video[0, :, t].transpose(1, 2, 0)))
            res = np.asarray(img.resize((224, 224), resample=PIL.BICUBIC))
            ref = np.stack([lut[c][res[:, :, c]] for c in range(3)])
            assert np.array_equal(got[0, :, t], ref)
    try:
        from transformers import BlipImageProcessor
    except Exception:  # pragma: no cover
        return
    ip = BlipImageProcessor(size={"height": 224, "width": 224})
    if "Pil" not in type(ip).__name__ and getattr(ip, "resample", 3) != 3:  # pragma: no cover
        return
    video = rng.integers(0, 256, (2, 3, 1, 120, 160), dtype=np.uint8)
    ref = ip(images=[v[:, 0] for v in video], return_tensors="np").pixel_values
    assert np.array_equal(orc.process_frames(video, size=224)[:, :, 0], ref)


def test_normalize_lut_is_the_hf_float_pipeline():
    lut = preprocess.normalize_lut()
    x = np.arange(256, dtype=np.uint8)
    img = np.broadcast_to(x[None, :, None], (3, 256, 1)).copy()          # (C, H, W) like the HF numpy path
    t = (img * (1 / 255)).astype(np.float32)                             # rescale(): uint8 * float -> float64 -> float32
    mean = np.asarray(preprocess.CLIP_MEAN, np.float32)[:, None, None]
    std = np.asarray(preprocess.CLIP_STD, np.float32)[:, None, None]
    assert np.array_equal(lut, ((t - mean) / std)[:, :, 0])


def test_process_frames_requires_gpu_tensor():
    import torch

    with pytest.raises(RuntimeError):
        preprocess.process_frames(torch.zeros((1, 3, 1, 8, 8), dtype=torch.uint8))
    with pytest.raises(ValueError):
        preprocess.process_frames(torch.zeros((1, 3, 1, 8, 8), dtype=torch.float32))


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("b,t,h,w,size", [(2, 3, 37, 53, 24), (1, 2, 224, 224, 224), (1, 1, 480, 640, 224), (1, 2, 100, 150, 224),
                                          (1, 1, 224, 300, 224), (1, 1, 360, 224, 224), (1, 8, 256, 341, 224)])
def test_hip_equals_oracle_bit_exact(b, t, h, w, size):
    import torch

    rng = np.random.default_rng(b * 1000 + h)
    video = rng.integers(0, 256, (b, 3, t, h, w), dtype=np.uint8)
    ref = orc.process_frames(video, size=size)
    dv = torch.from_numpy(video).cuda()
    got = preprocess.process_frames(dv, size=size)
    assert got.dtype == torch.float32 and tuple(got.shape) == ref.shape
    assert np.array_equal(got.cpu().numpy(), ref)
    got16 = preprocess.process_frames(dv, size=size, dtype=torch.bfloat16)
    assert torch.equal(got16, got.to(torch.bfloat16))  # bf16 output = RNE of the fp32 values (what model.to(bf16) does)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_equals_reference_golden(case):
    import torch

    g = np.load(GOLD)
    got = preprocess.process_frames(torch.from_numpy(g[f"{case}_video"]).cuda(), size=int(g[f"{case}_size"]))
    assert np.array_equal(got.cpu().numpy(), g[f"{case}_pixel_values"])


@pytest.mark.gpu
def test_process_routes_gpu_uint8_video_to_the_device_path():
    import torch

    from eilev.model.utils import process

    class _IP:
        size = {"height": 224, "width": 224}
        do_resize = do_rescale = do_normalize = True
        resample = 3
        image_mean, image_std, rescale_factor = preprocess.CLIP_MEAN, preprocess.CLIP_STD, 1 / 255

    class _Proc:
        image_processor = _IP()

        def __call__(self, text=None, return_tensors=None, **kw):
            from transformers import BatchEncoding

            return BatchEncoding({"input_ids": torch.tensor([[2, 5]])})

    video = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (3, 4, 120, 160), dtype=np.uint8)).cuda()  # (C, T, H, W)
    out = process(_Proc(), video=video, text="a")
    assert tuple(out["pixel_values"].shape) == (1, 3, 4, 224, 224) and out["pixel_values"].is_cuda
    assert np.array_equal(out["pixel_values"].cpu().numpy(), orc.process_frames(video[None].cpu().numpy(), size=224))
    assert "input_ids" in out
