#!/usr/bin/env python3
"""Host-inclusive wall-clock of Sampler.generate_seeds: seeds in, uint8 images on the host out -- per-image
torch.Generator noise (x_T and every z_t), PCIe upload, T steps, de-normalisation, download.  bench.py's `value`
starts with the noise resident in HBM; this is the figure DESIGN.md quotes beside it.

    python tools/e2e_generate.py [--batch 64] [--T 1000] [--size 64] [--upfront]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from synt_isic_amd.sampler import Sampler, draw_noise, run_sampling_loop  # noqa: E402
from synt_isic_amd.weights import synthetic_unet_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--upfront", action="store_true", help="also time the draw-everything-first path")
    a = ap.parse_args()
    s = Sampler()
    s.add_model("NV", synthetic_unet_state_dict())
    seeds = list(range(a.batch))
    s.generate_seeds("NV", seeds[:a.batch], 8, (a.size, a.size))          # warm-up: workspace, pinned buffers
    t0 = time.perf_counter()
    res = s.generate_seeds("NV", seeds, a.T, (a.size, a.size))
    img = res.images.cpu().numpy()
    dt = time.perf_counter() - t0
    print(f"streamed noise : {dt:.2f} s for {a.batch} images at {a.size}x{a.size}, T={a.T} -> {a.batch / dt:.3f} images/s "
          f"(host-inclusive), checksum {int(img.sum())}", flush=True)
    if a.upfront:
        t0 = time.perf_counter()
        sched = s.create_scheduler(a.T)
        x_T, z = draw_noise(seeds, a.T - 1, (3, a.size, a.size))
        t1 = time.perf_counter()
        r2 = run_sampling_loop(s.models["NV"], sched, x_T.to("cuda"), z.to("cuda"))
        img2 = r2.images.cpu().numpy()
        dt2 = time.perf_counter() - t0
        print(f"up-front noise : {dt2:.2f} s ({t1 - t0:.2f} s of it drawing) -> {a.batch / dt2:.3f} images/s, "
              f"identical images: {bool((img == img2).all())}", flush=True)


if __name__ == "__main__":
    main()
