#!/usr/bin/env python
"""Golden vectors for eilev_process_frames: uint8 frames -> pixel_values through the HF image processor the reference's
process() wraps (ref:eilev/model/utils.py:5-26; BlipImageProcessor, PIL backend, BICUBIC), written to
tests/golden/process_frames.npz.  Small sizes (the arithmetic does not depend on them); the fixture records the versions."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    import PIL
    import transformers
    from transformers import BlipImageProcessor

    rng = np.random.default_rng(20240928)
    cases = {}
    for name, (b, t, h, w, size) in {"down": (1, 2, 61, 83, 24), "up": (2, 1, 9, 13, 24), "wide_only": (1, 1, 24, 50, 24),
                                     "identity": (1, 2, 24, 24, 24), "tall_only": (1, 1, 40, 24, 24)}.items():
        video = rng.integers(0, 256, (b, 3, t, h, w), dtype=np.uint8)
        ip = BlipImageProcessor(size={"height": size, "width": size})
        frames = video.transpose(0, 2, 1, 3, 4).reshape(b * t, 3, h, w)            # process(): permute(0, 2, 1, 3, 4).flatten(end_dim=1)
        px = ip(images=[f for f in frames], return_tensors="np").pixel_values      # (b*t, 3, size, size) float32
        cases[f"{name}_video"] = video
        cases[f"{name}_pixel_values"] = px.reshape(b, t, 3, size, size).transpose(0, 2, 1, 3, 4).copy()
        cases[f"{name}_size"] = np.int64(size)
    cases["versions"] = np.array([f"transformers {transformers.__version__}", f"Pillow {PIL.__version__}", f"numpy {np.__version__}",
                                  f"image processor {type(ip).__name__}"])
    out = os.path.join(ROOT, "tests", "golden", "process_frames.npz")
    np.savez_compressed(out, **cases)
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    sys.exit(main())
