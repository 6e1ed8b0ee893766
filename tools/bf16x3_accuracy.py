#!/usr/bin/env python3
"""How accurate would an fp32-equivalent product on the bf16 matrix pipe be?  (NOT built: DESIGN.md section 8.)

Each fp32 operand is split exactly into three bf16 terms (8 + 8 + 8 significant bits); a product a * b is replaced by the
six term products with index sum <= 4 (hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi), each exact in fp32, accumulated in
fp32 -- what v_mfma_f32_32x32x16_bf16 would do at 16x the f32 MFMA rate, i.e. 2.7x the matrix throughput of
v_mfma_f32_32x32x2_f32.  This script emulates both accumulations in numpy for a contraction over K channels of O(1)
operands and reports the error against float64, in units of the largest result -- the quantity the per-kernel parity bound
(1e-5 * max(1, |ref|)) is stated in.    python tools/bf16x3_accuracy.py
"""
import numpy as np


def to_bf16(x):
    """round-to-nearest-even truncation of fp32 to bf16, returned as fp32"""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    hi = to_bf16(x)
    r1 = (x - hi).astype(np.float32)
    mid = to_bf16(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = to_bf16(r2)
    return hi, mid, lo, (r2 - lo).astype(np.float32)


def main():
    rng = np.random.default_rng(0)
    print(f"{'K':>6s} {'fp32 chain':>12s} {'bf16x3, 6 products':>20s} {'bf16x3, 9 products':>20s} {'split residue':>14s}")
    for K in (64, 192, 512, 1024, 4608):
        a = rng.standard_normal((64, K)).astype(np.float32) * np.float32(K ** -0.5)
        b = rng.standard_normal((K, 256)).astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        scale = max(1.0, np.abs(ref).max())
        # fp32 FMA chain in channel order (the f32 MFMA)
        acc = np.zeros((64, 256), np.float32)
        for k in range(K):
            acc = (acc.astype(np.float64) + a[:, k:k + 1].astype(np.float64) * b[k:k + 1, :].astype(np.float64)).astype(np.float32)
        e32 = np.abs(acc - ref).max() / scale
        ah, am, al, ares = split3(a)
        bh, bm, bl, bres = split3(b)
        def run(terms):
            acc = np.zeros((64, 256), np.float32)
            # the MFMA sums 16 products (8 channels x 2 terms) in higher precision and rounds once into the fp32 accumulator
            for k0 in range(0, K, 8):
                for (x, y), (x2, y2) in terms:
                    blockp = x[:, k0:k0 + 8].astype(np.float64) @ y[k0:k0 + 8, :].astype(np.float64)
                    blockp += x2[:, k0:k0 + 8].astype(np.float64) @ y2[k0:k0 + 8, :].astype(np.float64)
                    acc = (acc.astype(np.float64) + blockp).astype(np.float32)
            return np.abs(acc - ref).max() / scale
        e6 = run([((ah, bh), (ah, bm)), ((am, bh), (am, bm)), ((ah, bl), (al, bh))])
        z = np.zeros_like
        e9 = run([((ah, bh), (ah, bm)), ((am, bh), (am, bm)), ((ah, bl), (al, bh)), ((am, bl), (al, bm)), ((al, bl), (z(al), z(bl)))])
        print(f"{K:6d} {e32:12.3e} {e6:20.3e} {e9:20.3e} {max(np.abs(ares).max(), np.abs(bres).max()):14.1e}")


if __name__ == "__main__":
    main()
