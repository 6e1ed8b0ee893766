#!/usr/bin/env python3
"""Large / unusual shapes through the sampler: no crash, finite latents, batch independence of image 0.
    python tools/stress_shapes.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from synt_isic_amd.sampler import Sampler  # noqa: E402
from synt_isic_amd.weights import synthetic_unet_state_dict  # noqa: E402


def main():
    s = Sampler()
    s.add_model("NV", synthetic_unet_state_dict())
    ref = {}
    for B, size, T in [(1, 64, 3), (256, 64, 3), (1, 128, 3), (96, 128, 3), (1, 256, 2), (9, 256, 2), (1, 512, 1), (2, 384, 1)]:
        t0 = time.perf_counter()
        res = s.generate_seeds("NV", list(range(B)), T, (size, size))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        lat = res.latents
        ok = bool(torch.isfinite(lat).all())
        key = (size, T)
        same = ""
        if B == 1:
            ref[key] = lat[0].clone()
        elif key in ref:
            same = f", image 0 identical to the B=1 run: {bool(torch.equal(lat[0], ref[key]))}"
        print(f"B={B:4d} {size}x{size} T={T}: {dt:6.2f} s, finite={ok}, |x|max={float(lat.abs().max()):.3f}{same}", flush=True)


if __name__ == "__main__":
    main()
