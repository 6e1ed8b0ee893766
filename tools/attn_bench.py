#!/usr/bin/env python3
"""Times the attention core (q, k, v -> o for 32 heads x d = 8) at the token counts of the reference configurations, through the
C ABI.  Tuning aid (SISIC_ATT_PV_MFMA=0|1, SISIC_ATT_QB=1|2 select the forms); not a test.

    python tools/attn_bench.py [--iters 30]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from synt_isic_amd import ops  # noqa: E402

# (name, batch, channels, tokens)
CASES = [("64x64 images: 16x16 tokens, B=64", 64, 256, 256), ("128x128 images: 32x32 tokens, B=16", 16, 256, 1024),
         ("one image, 16x16 tokens", 1, 256, 256), ("ragged 15x15 tokens, B=8", 8, 256, 225)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    for name, B, C, N in CASES:
        qkv = torch.randn(B, 3 * C, N, device="cuda")
        ops.attention(qkv, 8)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.attention(qkv, 8)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        print(f"{name:40s} {us:9.1f} us   {4.0 * B * C * N * N / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
