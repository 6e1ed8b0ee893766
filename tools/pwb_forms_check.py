#!/usr/bin/env python3
"""Are the two forms of the bf16x3 pointwise kernel (tile_cfg 29: 32 pixels per wave, 30: 64) bit-equal in outputs and
GroupNorm partials?  A sweep over shapes and fused features; prints what differs."""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from synt_isic_amd import ops  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator().manual_seed(7)
bad = 0
for (B, c0, c1, cout, H) in [(2, 64, 0, 64, 16), (2, 128, 64, 64, 32), (1, 256, 0, 768, 16), (2, 256, 256, 256, 8), (3, 64, 64, 128, 16), (2, 40, 24, 192, 8)]:
    x = torch.randn(B, c0, H, H, generator=g).to(dev)
    x2 = torch.randn(B, c1, H, H, generator=g).to(dev) if c1 else None
    w = (torch.randn(cout, c0 + c1, 1, 1, generator=g) * (c0 + c1) ** -0.5).to(dev)
    wp = ops.pack_conv_weight(w)
    b = torch.randn(cout, generator=g).to(dev)
    res = torch.randn(B, cout, H, H, generator=g).to(dev)
    cb = torch.randn(B, cout, generator=g).to(dev)
    gs, gb = (1 + 0.3 * torch.randn(B, c0 + c1, generator=g)).to(dev), (0.3 * torch.randn(B, c0 + c1, generator=g)).to(dev)
    for use_b, use_res, use_cb, gn, relu in itertools.product((0, 1), (0, 1), (0, 1), (0, 1, 2), (0, 1)):
        kw = dict(bias=b if use_b else None, x2=x2, residual=res if use_res else None, chan_bias=cb if use_cb else None,
                  gn_scale=gs if gn else None, gn_shift=gb if gn else None, gn_silu=(gn == 2), relu=bool(relu), with_stats=True)
        y1, s1 = ops.conv2d(x, wp, cout, 1, tile_cfg=29, **kw)
        y2, s2 = ops.conv2d(x, wp, cout, 1, tile_cfg=30, **kw)
        ey, es = torch.equal(y1, y2), torch.equal(s1, s2)
        if not (ey and es):
            bad += 1
            dy = (y1 - y2).abs().max().item()
            ds = (s1 - s2).abs().max().item()
            print(f"shape {(B, c0, c1, cout, H)} bias {use_b} res {use_res} chan_bias {use_cb} gn {gn} relu {relu}: out equal {ey} ({dy:.3e}) stats equal {es} ({ds:.3e})")
print("differing cases:", bad)
