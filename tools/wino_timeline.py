#!/usr/bin/env python3
"""Timeline of one Winograd launch from the s_memtime stamps of the diagnostic build (make -C synt_isic_amd/csrc timing):
when each workgroup started and ended, on which CU, and how the phases of the workgroups that SHARE a CU lie against
each other.  Answers: is the launch memory time + matrix time (partners in lock-step) or their maximum?

    SISIC_LIB_PATH=tools/bin/libsisic_hip_timing.so python tools/wino_timeline.py [--cfg 71] [--cin 64] [--cout 64] [--hw 64]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from synt_isic_amd import _lib, ops  # noqa: E402
from synt_isic_amd._lib import ConvArgs, check  # noqa: E402

SLOTS = 10


def run(cfg, cin, cout, hw, B, res=True, clock_ghz=2.1, verbose_cus=2):
    dev = torch.device("cuda")
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, cin, hw, hw, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev)
    wp, ww = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)
    sc, sh = torch.ones(B, cin, device=dev), torch.zeros(B, cin, device=dev)
    r = torch.randn(B, cout, hw, hw, generator=g).to(dev)
    out = torch.empty(B, cout, hw, hw, device=dev)
    a = ConvArgs()
    a.in0 = x.data_ptr(); a.c0 = cin; a.B = B; a.Hin = hw; a.Win = hw; a.ksize = 3; a.stride = 1
    a.w_packed = wp.data_ptr(); a.Cout = cout; a.out = out.data_ptr(); a.w_winograd = ww.data_ptr()
    a.gn_scale = sc.data_ptr(); a.gn_shift = sh.data_ptr(); a.gn_silu = 1
    if res:
        a.residual = r.data_ptr()
    a.tile_cfg = cfg
    co_t = 128 if cfg in (68, 70, 72) else 64
    nwg = B * ((hw + 7) // 8) * ((hw + 15) // 16) * ((cout + co_t - 1) // co_t)
    stamps = torch.zeros(nwg * SLOTS + 64, dtype=torch.int64, device=dev)
    a.stats_out = stamps.data_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        e0.record()
        check(lib.sisic_conv2d(ops.context(dev), C.byref(a), None))
        e1.record()
    torch.cuda.synchronize()
    t = stamps[: nwg * SLOTS].view(nwg, SLOTS).cpu()
    hw_id, xcc = t[:, 8], t[:, 9]
    cu = ((xcc & 0xf) << 16) | (((hw_id >> 13) & 7) << 8) | (((hw_id >> 12) & 1) << 4) | ((hw_id >> 8) & 0xf)   # (xcc, se, sh, cu)
    ts = t[:, :8].double()
    t0 = ts[:, 0].min()
    us = (ts - t0) / (clock_ghz * 1e3)
    span = us[:, 7].max().item()
    print(f"=== cfg {cfg}: {cin}->{cout} @{hw}x{hw} B={B}: {nwg} workgroups, event time {e0.elapsed_time(e1) * 1e3:.1f} us, "
          f"first start -> last end {span:.1f} us (at {clock_ghz} GHz; {span * clock_ghz * 1e3:.0f} s_memtime ticks), "
          f"distinct CUs {len(set(cu.tolist()))}")
    life = us[:, 7] - us[:, 0]
    loop = us[:, 4] - us[:, 3]
    pro = us[:, 3] - us[:, 0]
    epi = us[:, 7] - us[:, 4]
    print(f"    per workgroup (median): life {life.median():.2f}  prologue {pro.median():.2f}  channel loop {loop.median():.2f}  "
          f"epilogue {epi.median():.2f}")
    starts = us[:, 0]
    print(f"    starts: p0 {starts.min():.1f}  p25 {starts.quantile(0.25):.1f}  p50 {starts.median():.1f}  p75 {starts.quantile(0.75):.1f}  "
          f"p100 {starts.max():.1f};  ends: p0 {us[:, 7].min():.1f} p50 {us[:, 7].median():.1f} p100 {us[:, 7].max():.1f}")
    # per CU: fraction of the span during which >= 1 / >= 2 workgroups are inside their channel loop
    by_cu = {}
    for i, c in enumerate(cu.tolist()):
        by_cu.setdefault(c, []).append(i)
    import numpy as np
    grid = np.linspace(0.0, span, 2000)
    in_loop_1 = in_loop_2 = resident_2 = 0.0
    for c, idx in by_cu.items():
        lo = np.zeros_like(grid)
        rs = np.zeros_like(grid)
        for i in idx:
            lo += (grid >= us[i, 3].item()) & (grid < us[i, 4].item())
            rs += (grid >= us[i, 0].item()) & (grid < us[i, 7].item())
        in_loop_1 += (lo >= 1).mean()
        in_loop_2 += (lo >= 2).mean()
        resident_2 += (rs >= 2).mean()
    n = len(by_cu)
    print(f"    per CU, share of the span: >= 1 workgroup in its channel loop {in_loop_1 / n:.2f}, two at once {in_loop_2 / n:.2f}; "
          f"two resident {resident_2 / n:.2f}; workgroups per CU {nwg / n:.1f}")
    for c in sorted(by_cu)[:verbose_cus]:
        print(f"    CU {c:#x}:")
        for i in sorted(by_cu[c], key=lambda i: us[i, 0].item()):
            u = us[i]
            print(f"        wg {i:5d}  start {u[0]:6.1f}  loop {u[3]:6.1f}-{u[4]:6.1f}  end {u[7]:6.1f}   wave slot {int(hw_id[i]) & 0xf}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=71)
    ap.add_argument("--cin", type=int, default=64)
    ap.add_argument("--cout", type=int, default=64)
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--all", action="store_true")
    args = ap.parse_args()
    if args.all:
        run(71, 8, 64, 64, 64)
        run(71, 64, 64, 64, 64)
        run(71, 128, 64, 64, 64, verbose_cus=0)
        run(70, 128, 128, 32, 64)
        run(70, 256, 256, 16, 64, verbose_cus=0)
    else:
        run(args.cfg, args.cin, args.cout, args.hw, args.batch)
