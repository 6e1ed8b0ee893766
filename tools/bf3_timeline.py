#!/usr/bin/env python3
"""Per-wave timeline of the bf16x3 Winograd kernel (tile_cfg 74) from the s_memtime stamps of the diagnostic build
(make -C synt_isic_amd/csrc timing): where a workgroup's time goes -- prologue, one chunk of the channel loop in detail
(first tile block issued and the next chunk staged, second block issued, barrier passed), the output rounds -- and how far the 16 waves of a
workgroup are apart at each point.

    SISIC_LIB_PATH=tools/bin/libsisic_hip_timing.so python tools/bf3_timeline.py [--cin 64] [--cout 64] [--hw 64]
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

SLOTS = 16
NAMES = {0: "start", 1: "prologue done", 2: "chunk 2 start", 3: "chunk 2 block 0 + staging", 4: "chunk 2 block 1 + loads", 5: "chunk 2 barrier",
         6: "chunk 3 start", 7: "chunk 3 block 0 + staging", 8: "chunk 3 block 1 + loads", 9: "chunk 3 barrier", 10: "loop done",
         11: "round 0 operands requested", 12: "round 0 planes written", 13: "round 0 exchanged", 14: "round 0 stored", 15: "end"}


def run(cin, cout, hw, B, res=True, clock_ghz=0.1):
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
    a.tile_cfg = 74
    nwg = B * ((hw + 15) // 16) ** 2 * ((cout + 63) // 64)
    stamps = torch.zeros(nwg * 16 * SLOTS + 64, dtype=torch.int64, device=dev)
    a.stats_out = stamps.data_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        e0.record()
        check(lib.sisic_conv2d(ops.context(dev), C.byref(a), None))
        e1.record()
    torch.cuda.synchronize()
    t = stamps[: nwg * 16 * SLOTS].view(nwg, 16, SLOTS).cpu().double()
    if os.environ.get("BF3_TIMELINE_RAW"):
        print(stamps[: 2 * SLOTS].tolist())
    zeros = [(k, int((t[:, :, k] == 0).sum())) for k in range(SLOTS) if (t[:, :, k] == 0).any()]
    if zeros:
        print("    unwritten stamps (slot, count):", zeros)
    # s_memtime is per XCD (the bases differ): only differences inside a workgroup mean anything.  Ticks per microsecond from
    # the launch: its workgroups run back to back on their CU, so (workgroups per CU) x (median life) is about the event time
    life_ticks = (t[:, :, 15].max(dim=1).values - t[:, :, 0].min(dim=1).values)
    ev_us = e0.elapsed_time(e1) * 1e3
    span_ticks = (t[:, :, 15].max().item() - t[:, :, 0].min().item()) if False else life_ticks.median().item() * max(nwg / 256.0, 1.0)
    tick_us = ev_us / span_ticks
    print(f"=== {cin}->{cout} @{hw}x{hw} B={B}: {nwg} workgroups, event time {ev_us:.1f} us ~ {span_ticks:.0f} ticks "
          f"({1 / tick_us:.1f} ticks / us)")
    rel = (t - t[:, :, 0:1].min(dim=1, keepdim=True).values) * tick_us           # us since the workgroup's first wave started
    print("    per workgroup, microseconds since its first wave started: median over workgroups of the FIRST / MEDIAN / LAST wave")
    prev = None
    for k in range(SLOTS):
        col = rel[:, :, k]
        if (t[:, :, k] == 0).all():
            continue
        f, m, l = col.min(dim=1).values.median().item(), col.median(dim=1).values.median().item(), col.max(dim=1).values.median().item()
        d = "" if prev is None else f"   (+{m - prev:5.2f})"
        print(f"      {NAMES[k]:26s} {f:7.2f} {m:7.2f} {l:7.2f}{d}")
        prev = m
    if os.environ.get("BF3_TIMELINE_PER_WAVE"):
        # is it always the same wave that arrives last?  Per wave (= Winograd position): median over workgroups of its time at a
        # stamp minus the workgroup's median wave at that stamp, and how often it is the workgroup's last wave there
        for k in (3, 4, 7, 8):
            col = rel[:, :, k]
            off = (col - col.median(dim=1, keepdim=True).values).median(dim=0).values
            last = torch.nn.functional.one_hot(col.argmax(dim=1), 16).double().mean(dim=0)
            print(f"    {NAMES[k]}: per wave, us after the workgroup's median wave / share of workgroups where it is last")
            print("      " + " ".join(f"{off[w]:+5.2f}" for w in range(16)))
            print("      " + " ".join(f"{last[w]:5.2f}" for w in range(16)))
    life = (t[:, :, 15].max(dim=1).values - t[:, :, 0].min(dim=1).values) * tick_us
    grid = min(nwg, 256)
    if nwg > grid and os.environ.get("SISIC_BF3_PERSISTENT", "1") != "0":
        # persistent workgroups: item L + grid follows item L on the same workgroup (same clock)
        first = life[:grid].median().item()
        later = life[grid:].median().item()
        gap = ((t[grid:, :, 0].min(dim=1).values - t[: nwg - grid, :, 15].max(dim=1).values) * tick_us).median().item()
        pro_first = rel[:grid, :, 1].median(dim=1).values.median().item()
        pro_later = rel[grid:, :, 1].median(dim=1).values.median().item()
        print(f"    persistent: first item {first:.2f} us (prologue {pro_first:.2f}), later items {later:.2f} us (prologue {pro_later:.2f}), "
              f"last wave's end -> first wave's next start {gap:.2f} us")
    print(f"    workgroup life: median {life.median():.2f} us; {nwg / 256:.1f} workgroups per CU -> {life.median() * nwg / 256:.1f} us if back to back")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=64)
    ap.add_argument("--cout", type=int, default=64)
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    run(args.cin, args.cout, args.hw, args.batch)
