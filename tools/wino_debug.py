#!/usr/bin/env python3
"""Where does a Winograd tile configuration differ from the float64 convolution?  Development aid, not a test.

Runs one 3x3 convolution through the C ABI with the given tile_cfg for a few channel counts, and with a single live input
channel at a time, and prints which input channels / output channels / pixels carry the error.

    python tools/wino_debug.py --cfg 76 [--cout 64] [--hw 16] [--batch 1]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from synt_isic_amd import ops  # noqa: E402


def run(x, w, cfg):
    d = lambda t: t.cuda().contiguous()
    return ops.conv2d(d(x), ops.pack_conv_weight(d(w)), w.shape[0], 3, tile_cfg=cfg, w_winograd=ops.pack_winograd_weight(d(w))).cpu().double()


def describe(err, tol):
    bad = err > tol
    if not bad.any():
        return "ok"
    b, co, y, x = [sorted(set(t.tolist())) for t in bad.nonzero(as_tuple=True)]
    short = lambda v: f"{v[0]}..{v[-1]} ({len(v)})" if len(v) > 6 else str(v)
    return f"max {err.max():.3e}; images {short(b)} out-channels {short(co)} rows {short(y)} cols {short(x)}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=76)
    ap.add_argument("--cout", type=int, default=64)
    ap.add_argument("--hw", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--cins", default="8,16,24,32,40,20")
    ap.add_argument("--onehot", type=int, default=32, help="channel count of the one-live-channel sweep (0 = skip)")
    a = ap.parse_args()
    g = torch.Generator().manual_seed(5)
    for cin in [int(v) for v in a.cins.split(",")]:
        x = torch.randn(a.batch, cin, a.hw, a.hw, generator=g)
        w = torch.randn(a.cout, cin, 3, 3, generator=g) * 0.1
        ref = F.conv2d(x.double(), w.double(), padding=1)
        err = (run(x, w, a.cfg) - ref).abs()
        print(f"cin {cin:4d}: {describe(err, 1e-5 * max(1.0, ref.abs().max().item()))}", flush=True)
    if a.onehot:
        cin = a.onehot
        w = torch.randn(a.cout, cin, 3, 3, generator=g) * 0.1
        for c in range(cin):
            x = torch.zeros(a.batch, cin, a.hw, a.hw)
            x[:, c] = torch.randn(a.batch, a.hw, a.hw, generator=g)
            ref = F.conv2d(x.double(), w.double(), padding=1)
            err = (run(x, w, a.cfg) - ref).abs()
            print(f"only channel {c:3d} of {cin}: {describe(err, 1e-5 * max(1.0, ref.abs().max().item()))}", flush=True)
        # one live pixel of channel 0: which outputs see it
        x = torch.zeros(a.batch, cin, a.hw, a.hw)
        x[:, 0, 5, 6] = 1.0
        ref = F.conv2d(x.double(), w.double(), padding=1)
        got = run(x, w, a.cfg)
        err = (got - ref).abs()
        print(f"one pixel (5,6) of channel 0: {describe(err, 1e-5)}")
        print("got[0,0,3:8,4:9]\n", got[0, 0, 3:8, 4:9], "\nref\n", ref[0, 0, 3:8, 4:9])
    # which of the six term products is off?  (channel 1 of 16 live)
    cin = 16
    bf = lambda t: t.bfloat16().float()
    wc = torch.zeros(a.cout, cin, 3, 3)
    wc[:, :, 1, 1] = bf(torch.randn(a.cout, cin, generator=g))           # U = G g G^T exact in bf16: U_mid = U_lo = 0
    wr = torch.randn(a.cout, cin, 3, 3, generator=g) * 0.1
    xr = torch.zeros(a.batch, cin, a.hw, a.hw)
    xr[:, 1] = torch.randn(a.batch, a.hw, a.hw, generator=g)
    xp = torch.zeros(a.batch, cin, a.hw, a.hw)
    xp[:, 1, 5, 6] = 1.0                                                 # V exact in bf16: V_mid = V_lo = 0
    xp[:, 1, 2, 9] = -0.5
    for name, x, w in (("U exact (tests U_hi x V_mid, V_lo)", xr, wc), ("V exact (tests U_mid, U_lo x V_hi)", xp, wr), ("both exact", xp, wc)):
        ref = F.conv2d(x.double(), w.double(), padding=1)
        got = run(x, w, a.cfg)
        err = (got - ref).abs()
        print(f"{name}: {describe(err, 1e-6 * max(1.0, ref.abs().max().item()))}   (|ref| max {ref.abs().max():.3f})")


if __name__ == "__main__":
    main()
