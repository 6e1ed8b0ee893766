#!/usr/bin/env python3
"""Times sisic_conv2d_wgrad on the UNet's 3x3 stride-1 layer shapes (batch 32 at 64x64 input).
    python tools/wgrad_bench.py [batch [layer]]        (SISIC_WGRAD_WINOGRAD=0 in the environment: the direct kernel)
The wall times include the entry point's scratch allocation (hundreds of microseconds); for kernel times run one layer under
rocprofv3:  bash tools/prof_script.sh w tools/wgrad_bench.py 32 <layer>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from synt_isic_amd import ops  # noqa: E402

LAYERS = [("64->64 @64", 64, 0, 64, 64), ("128+64->64 @64", 128, 64, 64, 64), ("64->128 @32", 64, 0, 128, 32),
          ("128->128 @32", 128, 0, 128, 32), ("256+128->128 @32", 256, 128, 128, 32), ("128->256 @16", 128, 0, 256, 16),
          ("256->256 @16", 256, 0, 256, 16), ("256+256->256 @16", 256, 256, 256, 16), ("256->256 @8", 256, 0, 256, 8),
          ("256+256->256 @8", 256, 256, 256, 8)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    only = int(sys.argv[2]) if len(sys.argv) > 2 else -1          # one layer (for rocprofv3: per-kernel statistics of one shape)
    dev = torch.device("cuda")
    print(f"SISIC_WGRAD_WINOGRAD={os.environ.get('SISIC_WGRAD_WINOGRAD', '1')}  B={B}")
    for idx, (name, c0, c1, cout, H) in enumerate(LAYERS):
        if only >= 0 and idx != only:
            continue
        x = torch.randn(B, c0, H, H, device=dev)
        x2 = torch.randn(B, c1, H, H, device=dev) if c1 else None
        dy = torch.randn(B, cout, H, H, device=dev)
        gs = torch.rand(B, c0 + c1, device=dev) + 0.5
        gb = torch.randn(B, c0 + c1, device=dev) * 0.1
        run = lambda: ops.conv2d_wgrad(x, dy, 3, x2=x2, gn_scale=gs, gn_shift=gb, gn_silu=True)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        flops = 2.0 * 9 * B * cout * (c0 + c1) * H * H
        print(f"{name:20s} {us:8.1f} us  {flops / us / 1e6:7.1f} direct-form TFLOP/s")


if __name__ == "__main__":
    main()
