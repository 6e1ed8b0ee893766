#!/usr/bin/env python3
"""Where a Winograd workgroup's time goes: reads the s_memtime stamps of a diagnostic build (-DWINO_TIMING=1, see
conv_winograd.hip) and prints the median duration of each phase in microseconds.

    hipcc ... -DWINO_TIMING=1 ... -o tools/bin/libsisic_hip_timing.so      (same sources, diagnostic define)
    SISIC_LIB_PATH=tools/bin/libsisic_hip_timing.so python tools/wino_phases.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ctypes as C  # noqa: E402

import torch  # noqa: E402

from synt_isic_amd import _lib, ops  # noqa: E402
from synt_isic_amd._lib import ConvArgs, check  # noqa: E402

PHASES = ["index plans", "first chunk loaded+staged", "transform(0), stage(1)", "channel loop", "epilogue operand requests",
          "output transform", "stores retired"]


def run(cin, cout, hw, B=64, gn=True, res=True, clock_ghz=2.4, cfg=66):
    dev = torch.device("cuda")
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, cin, hw, hw, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev)
    wp, ww = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)
    sc = torch.ones(B, cin, device=dev)
    sh = torch.zeros(B, cin, device=dev)
    r = torch.randn(B, cout, hw, hw, generator=g).to(dev)
    out = torch.empty(B, cout, hw, hw, device=dev)
    a = ConvArgs()
    a.in0 = x.data_ptr(); a.c0 = cin; a.B = B; a.Hin = hw; a.Win = hw; a.ksize = 3; a.stride = 1
    a.w_packed = wp.data_ptr(); a.Cout = cout; a.out = out.data_ptr(); a.w_winograd = ww.data_ptr()
    if gn:
        a.gn_scale = sc.data_ptr(); a.gn_shift = sh.data_ptr(); a.gn_silu = 1
    if res:
        a.residual = r.data_ptr()
    a.tile_cfg = cfg
    if cfg in (68, 70, 72):
        nwg = B * ((hw + 7) // 8) * ((hw + 15) // 16) * ((cout + 127) // 128)
    elif cfg in (69, 71, 73):
        nwg = B * ((hw + 7) // 8) * ((hw + 15) // 16) * ((cout + 63) // 64)
    else:
        nwg = B * ((hw + 15) // 16) ** 2 * ((cout + 63) // 64)
    stamps = torch.zeros(nwg * 10 + 64, dtype=torch.int64, device=dev)
    a.stats_out = stamps.data_ptr()
    for _ in range(3):
        check(lib.sisic_conv2d(ops.context(dev), C.byref(a), None))
    torch.cuda.synchronize()
    t = stamps[: nwg * 10].view(nwg, 10)[:, :8].cpu().double()
    d = (t[:, 1:] - t[:, :-1]) / (clock_ghz * 1e3)
    life = (t[:, 7] - t[:, 0]) / (clock_ghz * 1e3)
    print(f"--- cfg {cfg}: {cin}->{cout} @{hw}x{hw} B={B} ({nwg} workgroups, {cin // 8} chunks), median us per workgroup "
          f"(assuming {clock_ghz} GHz); workgroup life {life.median():.2f}")
    for name, col in zip(PHASES, d.t()):
        print(f"   {name:28s} {col.median():7.2f}   (p10 {col.quantile(0.1):6.2f}, p90 {col.quantile(0.9):6.2f})")


if __name__ == "__main__":
    for cfg in (66, 69):
        run(8, 64, 64, cfg=cfg)
        run(64, 64, 64, cfg=cfg)
        run(128, 64, 64, cfg=cfg)
    for cfg in (66, 68):
        run(128, 128, 32, cfg=cfg)
        run(256, 256, 16, cfg=cfg)
        run(512, 256, 16, cfg=cfg)
