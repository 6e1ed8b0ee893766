#!/usr/bin/env python3
"""Single-image latency (the reference's own use: generate_single_image, B=1): ms per denoising step and seconds per
image for T=50 at 64x64 and 128x128.   python tools/latency_b1.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from synt_isic_amd.sampler import Sampler  # noqa: E402
from synt_isic_amd.weights import synthetic_unet_state_dict  # noqa: E402


def main():
  for latency in (False, True):
    s = Sampler(latency_mode=latency)
    s.add_model("NV", synthetic_unet_state_dict())
    print(f"--- latency_mode={latency}", flush=True)
    for size in (64, 128):
        for B in (1, 4):
            s.generate_seeds("NV", list(range(B)), 8, (size, size))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = s.generate_seeds("NV", list(range(B)), 50, (size, size))
            img = res.images.cpu()
            dt = time.perf_counter() - t0
            print(f"B={B} {size}x{size} T=50: {dt:.3f} s per call, {dt / 50 * 1e3:.2f} ms per step, "
                  f"{B / dt:.2f} images/s", flush=True)


if __name__ == "__main__":
    main()
