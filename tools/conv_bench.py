#!/usr/bin/env python3
"""Times the distinct convolution instances of the reference UNet (B=64, 3x64x64 input) per tile
configuration through the C ABI.  Tuning aid for the dispatch table in conv_mfma.hip; not a test.

    python tools/conv_bench.py [--batch 64] [--iters 20] [--cfgs all|auto]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from synt_isic_amd import ops  # noqa: E402

# (name, count per forward, c0, c1, cout, H(in), ksize, stride, upsample, gn, residual)
LAYERS = [
    ("conv_in 3->64 @64", 1, 3, 0, 64, 64, 3, 1, 0, 0, 0),
    ("64->64 @64 gn", 4, 64, 0, 64, 64, 3, 1, 0, 1, 0),
    ("64->64 @64 gn+res", 5, 64, 0, 64, 64, 3, 1, 0, 1, 1),
    ("128+64->64 @64 gn", 1, 128, 64, 64, 64, 3, 1, 0, 1, 0),
    ("64+64->64 @64 gn", 2, 64, 64, 64, 64, 3, 1, 0, 1, 0),
    ("up 128->128 32->64", 1, 128, 0, 128, 32, 3, 1, 1, 0, 0),
    ("conv_out 64->3 @64 gn", 1, 64, 0, 3, 64, 3, 1, 0, 1, 0),
    ("down 64->64 s2 64->32", 1, 64, 0, 64, 64, 3, 2, 0, 0, 0),
    ("64->128 @32 gn", 1, 64, 0, 128, 32, 3, 1, 0, 1, 0),
    ("128->128 @32 gn+res", 6, 128, 0, 128, 32, 3, 1, 0, 1, 1),
    ("256+128->128 @32 gn", 1, 256, 128, 128, 32, 3, 1, 0, 1, 0),
    ("128+128->128 @32 gn", 1, 128, 128, 128, 32, 3, 1, 0, 1, 0),
    ("128+64->128 @32 gn", 1, 128, 64, 128, 32, 3, 1, 0, 1, 0),
    ("up 256->256 16->32", 1, 256, 0, 256, 16, 3, 1, 1, 0, 0),
    ("down 128->128 s2 32->16", 1, 128, 0, 128, 32, 3, 2, 0, 0, 0),
    ("128->256 @16 gn", 1, 128, 0, 256, 16, 3, 1, 0, 1, 0),
    ("256->256 @16 gn+res", 6, 256, 0, 256, 16, 3, 1, 0, 1, 1),
    ("256+256->256 @16 gn", 2, 256, 256, 256, 16, 3, 1, 0, 1, 0),
    ("256+128->256 @16 gn", 1, 256, 128, 256, 16, 3, 1, 0, 1, 0),
    ("up 256->256 8->16", 1, 256, 0, 256, 8, 3, 1, 1, 0, 0),
    ("down 256->256 s2 16->8", 1, 256, 0, 256, 16, 3, 2, 0, 0, 0),
    ("256->256 @8 gn+res", 11, 256, 0, 256, 8, 3, 1, 0, 1, 1),
    ("256+256->256 @8 gn", 3, 256, 256, 256, 8, 3, 1, 0, 1, 0),
    ("1x1 qkv 256->768 @16 gn", 5, 256, 0, 768, 16, 1, 1, 0, 2, 0),
    ("1x1 out 256->256 @16 res", 5, 256, 0, 256, 16, 1, 1, 0, 0, 1),
    ("1x1 sc 256+256->256 @16", 2, 256, 256, 256, 16, 1, 1, 0, 0, 0),
    ("1x1 sc 256+128->128 @32", 1, 256, 128, 128, 32, 1, 1, 0, 0, 0),
    ("1x1 sc 128+64->64 @64", 1, 128, 64, 64, 64, 1, 1, 0, 0, 0),
    ("1x1 sc 256+256->256 @8", 3, 256, 256, 256, 8, 1, 1, 0, 0, 0),
    # not layers of the network (count 0): per-workgroup cost of a quarter / an eighth of the 8x8 level's channels
    ("probe 64->256 @8 gn+res", 0, 64, 0, 256, 8, 3, 1, 0, 1, 1),
    ("probe 128->256 @8 gn", 0, 128, 0, 256, 8, 3, 1, 0, 1, 0),
    # fixed cost vs per-chunk cost of the Winograd kernel: one to sixteen 8-channel chunks at 64x64
    ("probe 8->64 @64 gn+res", 0, 8, 0, 64, 64, 3, 1, 0, 1, 1),
    ("probe 16->64 @64 gn+res", 0, 16, 0, 64, 64, 3, 1, 0, 1, 1),
    ("probe 32->64 @64 gn+res", 0, 32, 0, 64, 64, 3, 1, 0, 1, 1),
    ("probe 128->64 @64 gn+res", 0, 128, 0, 64, 64, 3, 1, 0, 1, 1),
    # 1x1: fixed cost vs contraction length (the UNet's 1x1 convolutions contract over 192..512 channels only)
    ("probe1x1 128->256 @16", 0, 128, 0, 256, 16, 1, 1, 0, 0, 0),
    ("probe1x1 256->256 @16", 0, 256, 0, 256, 16, 1, 1, 0, 0, 0),
    ("probe1x1 512->256 @16", 0, 512, 0, 256, 16, 1, 1, 0, 0, 0),
    ("probe1x1 1024->256 @16", 0, 1024, 0, 256, 16, 1, 1, 0, 0, 0),
    ("probe1x1 2048->256 @16", 0, 2048, 0, 256, 16, 1, 1, 0, 0, 0),
]


def cfgs_for(k, stride, which):
    if which == "auto":
        return [0]
    if which != "all":
        return [int(c) for c in which.split(",")]
    if k == 1:
        return [0, 22, 23, 24, 25, 26, 27]
    if stride == 2:
        return [0, 11, 12, 13]
    return [4, 16, 17, 66, 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cfgs", default="all")
    ap.add_argument("--scale", type=int, default=1, help="multiply H (2 = the 128x128 configuration)")
    ap.add_argument("--match", default="", help="only layers whose name contains this substring")
    args = ap.parse_args()
    dev = torch.device("cuda")
    B = args.batch
    total_ms = 0.0
    total_flops = 0.0
    print(f"{'layer':34s} {'n':>2s} {'cfg':>3s} {'us':>9s} {'TFLOP/s':>8s} {'GB/s':>7s}")
    for (name, count, c0, c1, cout, H, k, stride, ups, gn, res) in LAYERS:
        if args.match and args.match not in name:
            continue
        H *= args.scale
        W = H
        x = torch.randn(B, c0, H, W, device=dev)
        x2 = torch.randn(B, c1, H, W, device=dev) if c1 else None
        w = torch.randn(cout, c0 + c1, k, k, device=dev) * 0.05
        wp = ops.pack_conv_weight(w)
        wino = ops.pack_winograd_weight(w) if (k == 3 and stride == 1 and cout > 4) else None
        bias = torch.randn(cout, device=dev)
        Hc = 2 * H if ups else H
        Ho = (Hc + 2 * (k // 2) - k) // stride + 1
        gs = torch.rand(B, c0 + c1, device=dev) + 0.5 if gn else None
        gb = torch.randn(B, c0 + c1, device=dev) * 0.1 if gn else None
        r = torch.randn(B, cout, Ho, Ho, device=dev) if res else None
        flops = 2.0 * B * cout * Ho * Ho * (c0 + c1) * k * k
        byts = 4.0 * (B * (c0 + c1) * H * W + B * cout * Ho * Ho * (2 if res else 1) + (c0 + c1) * cout * k * k)
        best = None
        for cfg in cfgs_for(k, stride, args.cfgs):
            def run():
                return ops.conv2d(x, wp, cout, k, bias=bias, x2=x2, stride=stride, upsample=bool(ups),
                                  gn_scale=gs, gn_shift=gb, gn_silu=(gn == 1), residual=r, tile_cfg=cfg,
                                  w_winograd=wino if (cfg == 0 or 60 <= cfg <= 79 or cfg in (90, 91, 92)) else None)
            try:
                run()
            except Exception as e:  # cfg not applicable
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            print(f"{name:34s} {count:2d} {cfg:3d} {us:9.1f} {flops / us / 1e6:8.1f} {byts / us / 1e3:7.0f}", flush=True)
            if cfg == 0 or "auto_us" not in dir():
                auto_us = us
            if best is None or us < best[1]:
                best = (cfg, us)
        if best is None:
            continue
        total_ms += count * auto_us / 1e3
        total_flops += count * flops
        if args.cfgs != "auto":
            print(f"{'':34s}    best cfg {best[0]} {best[1]:.1f} us (auto {auto_us:.1f})")
    print(f"sum over one forward (auto cfg): {total_ms:.3f} ms, {total_flops / max(total_ms, 1e-9) / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
