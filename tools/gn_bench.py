#!/usr/bin/env python3
"""Times gn_stats on the GroupNorm inputs of the UNet (B=64, 64x64).  Tuning aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synt_isic_amd import ops
dev = torch.device("cuda")
B = 64
tot = 0.0
for (c0, c1, r, n) in [(64, 0, 64, 10), (128, 64, 64, 1), (64, 64, 64, 2), (128, 0, 32, 7), (64, 0, 32, 1), (256, 128, 32, 1),
                       (128, 128, 32, 1), (128, 64, 32, 1), (256, 0, 16, 13), (128, 0, 16, 1), (256, 256, 16, 2), (256, 128, 16, 1),
                       (256, 0, 8, 11), (256, 256, 8, 3)]:
    x = torch.randn(B, c0, r, r, device=dev)
    x2 = torch.randn(B, c1, r, r, device=dev) if c1 else None
    g, b = torch.ones(c0 + c1, device=dev), torch.zeros(c0 + c1, device=dev)
    ops.groupnorm_stats(x, g, b, 32, 1e-5, x2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ops.groupnorm_stats(x, g, b, 32, 1e-5, x2)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    gb = 4.0 * B * (c0 + c1) * r * r / us / 1e3
    tot += n * us
    print(f"C={c0}+{c1} @{r}: {us:7.1f} us  {gb:7.0f} GB/s  x{n}")
print(f"sum per forward: {tot/1e3:.3f} ms")
