// pack_device.h -- one destination element of each weight re-layout, shared by the stand-alone pack kernels (one launch per
// tensor: load time) and pack_batch_kernel (repack.hip: every derived form of every weight in three launches after an
// optimizer step).  One definition, so the two routes cannot drift apart.
#pragma once
#include <hip/hip_runtime.h>

namespace sisic {

// OIHW -> [Cin_pad][KK][cout_pad], zero padded (conv_mfma.hip).  1x1 filters carry two more layouts behind the first
// (conv_packed_floats(), common.h: 3.5 times the elements):
//   second  [8-channel chunk][32-channel block][lane = (ci & 1) * 32 + co % 32][(ci % 8) / 2] -- the A fragments of
//           conv_pointwise.hip, one 16-byte load per lane and 8 input channels;
//   third   [8-channel chunk][64-channel tile][768 dwords] -- the filter split into three bf16 terms, the A operands of
//           conv_pointwise_bf3.hip: (hi, mid) of channel block 0 [64 lanes][4 dwords], of block 1, then lo of block 0
//           [64 lanes][2] and of block 1; lane (co = lane & 31, g = lane >> 5) carries input channels 4 g .. 4 g + 3 of the
//           chunk, a term's two dwords are channels (1 : 0) and (3 : 2) of the group (as winograd_pack_bf3_elem below).
__device__ __forceinline__ void conv_pack_elem(size_t i, const float* __restrict__ w, int Cout, int Cin, int KK, int cin_pad,
                                               int cout_pad, float* __restrict__ out) {
    const size_t first = (size_t)cin_pad * KK * cout_pad;
    if (i >= 2 * first) {               // (KK == 1 only) third layout
        const size_t j = i - 2 * first;
        const int n_co64 = cout_pad >> 6;
        const int wd = (int)(j % 768);
        const size_t r = j / 768;
        const int tile = (int)(r % n_co64), chunk = (int)(r / n_co64);
        int blk, ln, term, pair;
        if (wd < 512) { blk = wd >> 8; ln = (wd >> 2) & 63; term = (wd >> 1) & 1; pair = wd & 1; }      // 0 hi, 1 mid
        else { blk = (wd - 512) >> 7; ln = (wd >> 1) & 63; term = 2; pair = wd & 1; }                   // 2 lo
        const int co = 64 * tile + 32 * blk + (ln & 31), g = ln >> 5;
        unsigned pk = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ci = 8 * chunk + 4 * g + 2 * pair + e;
            const float v = (co < Cout && ci < Cin) ? w[(size_t)co * Cin + ci] : 0.0f;
            const unsigned hi = __float_as_uint(v) & 0xffff0000u;
            const float r1 = v - __uint_as_float(hi);
            const unsigned mid = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(mid);
            const unsigned t = term == 0 ? hi : (term == 1 ? mid : __float_as_uint(r2));
            pk |= (t >> 16) << (16 * e);
        }
        out[i] = __uint_as_float(pk);
        return;
    }
    if (i >= first) {                   // (KK == 1 only) second layout
        const size_t j = i - first;
        const int s = (int)(j & 3), ln = (int)((j >> 2) & 63);
        const size_t blk = j >> 8;
        const int n_co32 = cout_pad >> 5;
        const int mb = (int)(blk % n_co32), chunk = (int)(blk / n_co32);
        const int co2 = 32 * mb + (ln & 31), ci2 = 8 * chunk + 2 * s + (ln >> 5);
        out[i] = (co2 < Cout && ci2 < Cin) ? w[(size_t)co2 * Cin + ci2] : 0.0f;
        return;
    }
    const int co = (int)(i % cout_pad);
    const size_t r = i / cout_pad;
    const int tap = (int)(r % KK);
    const int ci = (int)(r / KK);
    float v = 0.0f;
    if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * KK + tap];
    out[i] = v;
}

// U = G g G^T of one (co, ci) pair, float64 arithmetic, written as [Cin_pad][16][cout_pad] (zero padded); i over cin_pad * cout_pad
__device__ __forceinline__ void winograd_pack_elem(size_t i, const float* __restrict__ w, int Cout, int Cin, int cin_pad,
                                                   int cout_pad, float* __restrict__ out) {
    const int co = (int)(i % cout_pad), ci = (int)(i / cout_pad);
    double u[4][4];
    if (co < Cout && ci < Cin) {
        const float* g = w + ((size_t)co * Cin + ci) * 9;
        double t[4][3];
        for (int j = 0; j < 3; ++j) {
            const double g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
            t[0][j] = g0;
            t[1][j] = 0.5 * (g0 + g1 + g2);
            t[2][j] = 0.5 * (g0 - g1 + g2);
            t[3][j] = g2;
        }
        for (int r = 0; r < 4; ++r) {
            u[r][0] = t[r][0];
            u[r][1] = 0.5 * (t[r][0] + t[r][1] + t[r][2]);
            u[r][2] = 0.5 * (t[r][0] - t[r][1] + t[r][2]);
            u[r][3] = t[r][2];
        }
    } else {
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) u[r][c] = 0.0;
    }
    for (int xi = 0; xi < 16; ++xi) out[((size_t)ci * 16 + xi) * cout_pad + co] = (float)u[xi >> 2][xi & 3];
}

// second packing of U for the wide form: [chunk][xi][32-channel block][lane = (ci & 1) * 32 + co % 32][(ci % 8) / 2];
// i over cin_pad * 16 * cout_pad128, read from the first layout [ci][xi][co]
__device__ __forceinline__ void winograd_pack_wide_elem(size_t i, const float* __restrict__ u_first, int cout_pad, int cout_pad128,
                                                        float* __restrict__ out) {
    const int n_co32 = cout_pad128 >> 5;
    const int co = (int)(i % cout_pad128);
    const size_t r = i / cout_pad128;
    const int xi = (int)(r % 16), ci = (int)(r / 16);
    const int chunk = ci >> 3, cp = (ci & 7) >> 1, hf = ci & 1;
    const size_t o = ((((size_t)chunk * 16 + xi) * n_co32 + (co >> 5)) * 64 + hf * 32 + (co & 31)) * 4 + cp;
    out[o] = co < cout_pad ? u_first[((size_t)ci * 16 + xi) * cout_pad + co] : 0.0f;
}

// third packing of U = G g G^T: the A operands of the bf16x3 form's MFMAs, [chunk of 8 ci][position][64-channel tile][768 dwords]:
// three 1 KB pieces, each one wave-wide 16-byte-per-lane transfer: (U_hi, U_mid) of channel block 0 [64 lanes][4], of block 1,
// then U_lo of block 0 [64 lanes][2] and of block 1.  Lane (co = lane & 31, g = lane >> 5) of a block carries input channels
// 4 g .. 4 g + 3 of the chunk; a term's two dwords are channels (1 : 0) and (3 : 2) of the group
__device__ __forceinline__ void winograd_pack_bf3_elem(size_t i, const float* __restrict__ u_first, int cout_pad, int cout_pad128,
                                                       unsigned* __restrict__ out) {
    const int n_co64 = cout_pad128 >> 6;
    const int w = (int)(i % 768);
    size_t r = i / 768;
    const int tile = (int)(r % n_co64); r /= n_co64;
    const int pos = (int)(r % 16);
    const int chunk = (int)(r / 16);
    int blk, ln, term, pair;
    if (w < 512) { blk = w >> 8; ln = (w >> 2) & 63; term = (w >> 1) & 1; pair = w & 1; }       // 0 hi, 1 mid
    else { blk = (w - 512) >> 7; ln = (w >> 1) & 63; term = 2; pair = w & 1; }                  // 2 lo
    const int co = 64 * tile + 32 * blk + (ln & 31), g = ln >> 5;
    unsigned pk = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int ci = 8 * chunk + 4 * g + 2 * pair + e;
        const float v = co < cout_pad ? u_first[((size_t)ci * 16 + pos) * cout_pad + co] : 0.0f;
        const unsigned hi = __float_as_uint(v) & 0xffff0000u;
        const float r1 = v - __uint_as_float(hi);
        const unsigned mid = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(mid);
        const unsigned t = term == 0 ? hi : (term == 1 ? mid : __float_as_uint(r2));
        pk |= (t >> 16) << (16 * e);
    }
    out[i] = pk;
}

// W'[ci][co][KK-1-t] = W[co][ci][t]: the filter of the backward-data convolution; i over Cout * Cin * KK
__device__ __forceinline__ void transpose_flip_elem(size_t i, const float* __restrict__ w, int Cout, int Cin, int KK,
                                                    float* __restrict__ wt) {
    const int t = (int)(i % KK);
    const size_t cc = i / KK;
    const int ci = (int)(cc % Cin), co = (int)(cc / Cin);
    wt[((size_t)ci * Cout + co) * KK + (KK - 1 - t)] = w[i];
}

// out[c][out_col0 + r] = in[r][c]; i over rows * cols
__device__ __forceinline__ void transpose2d_elem(size_t i, const float* __restrict__ in, int cols, float* __restrict__ out, int out_ld,
                                                 int out_col0) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    out[(size_t)c * out_ld + out_col0 + r] = in[i];
}

}  // namespace sisic
