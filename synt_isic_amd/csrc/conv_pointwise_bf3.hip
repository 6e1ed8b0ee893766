// conv_pointwise_bf3.hip -- 1x1 stride-1 convolutions with fp32-EQUIVALENT products on the bf16 matrix pipe (tile_cfg 28).
//
// The arithmetic of conv_winograd_bf3.inc applied to the plain GEMM D[co, px] = sum_ci W[co, ci] * act(X[ci, px]): every fp32
// operand is split exactly into three bf16 terms (hi + mid + lo, 8 + 8 + 8 significant bits), six of the nine term products --
// all but the three below 2^-24 of the product -- are summed by three v_mfma_f32_32x32x16_bf16 into the fp32 accumulator.
// Three 32-cycle instructions per 8 input channels replace the four 64-cycle f32 MFMAs of conv_pointwise.hip (tile_cfg 20);
// the measured error against float64 is the f32 kernels' (tests/test_gpu_kernels.py).
//
// A 1x1 convolution has no halo and no transform, so nothing is shared between waves that would be worth a barrier: a WAVE is
// the unit of work -- 64 (or 32) pixels x 64 output channels of one image -- and runs its whole contraction alone:
//   B  the lane's own loads: lane = (pixel pair 2 l, 2 l + 1; channel group g): one 8-byte load per channel and chunk of 8
//      channels gives the group's four channels for both 32-pixel blocks (block = pixel parity); GroupNorm (+ SiLU) and the
//      split happen in registers and ARE the operand -- with the K grouping of conv_winograd_bf3.inc nothing crosses lanes
//   A  the split filters straight from global memory in the third region of the packed 1x1 filter (pack_device.h): per chunk
//      and 32-channel block 16 bytes (hi, mid) + 8 bytes (lo) per lane
//   the activations one chunk ahead in registers; no LDS but the image's GroupNorm operands (a private table per wave), no
//   barrier anywhere; four waves per SIMD hide each other's latencies.  (Measured and not kept, profiles/r03/
//   conv_bench_pointwise_bf16x3_variants.txt: the filter chunk fetched once per 4-wave workgroup into LDS, activations four
//   chunks ahead -- within 3 % of this form on every layer.)
//   D  accumulators leave as 8-byte stores (the pixel pair) through a buffer resource: the channel is a scalar offset
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace sisic {

typedef float pwb_f32x16 __attribute__((ext_vector_type(16)));
typedef short pwb_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pwb_u4 __attribute__((ext_vector_type(4)));
typedef unsigned pwb_u2 __attribute__((ext_vector_type(2)));
typedef float pwb_f2 __attribute__((ext_vector_type(2)));
typedef float pwb_f4 __attribute__((ext_vector_type(4)));

struct PwbParams {
    const float* in0;
    const float* in1;
    int c0, c1, B, HW;
    const float* wb;         // [Cin/8][cout_pad/64][768 dwords] (pack_device.h, conv_pack_elem's third region)
    int n_co64;
    const float* bias;
    int Cout;
    const float* gn_scale;
    const float* gn_shift;
    const float* chan_bias;
    int chan_bias_stride;
    const float* residual;
    int relu;
    float* out;
    float* stats;            // optional [B][Cout][HW / 32][4]
    int n_px, n_co_items, nitems, nwg, nchunks;
};

constexpr int PWB_WAVES = 4;             // waves (= independent work items) per workgroup

__device__ __forceinline__ float pwb_half_wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
    return v + __shfl_xor(v, 16);
}
// the first four levels of that tree: the sum over a row of 16 lanes, in every lane of the row
__device__ __forceinline__ float pwb_row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ float pwb_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// PRO: 0 = no prologue, 1 = GroupNorm apply, 2 = + SiLU;  NB = 32-pixel blocks per wave: 2 = the lane's pixel PAIR (8-byte
// loads and stores, block = pixel parity), 1 = one pixel per lane
template <int PRO, int NB>
__global__ void __launch_bounds__(64 * PWB_WAVES, 4) conv_pwb_kernel(const PwbParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int wg;
    {   // XCD-aware bijective remap (conv_mfma.hip)
        const int L = blockIdx.x, nwg = p.nwg;
        const int xcd = L & 7, slot = L >> 3, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item = wg * PWB_WAVES + wave_u;
    if (item >= p.nitems) return;                                   // (wave-uniform; there is no barrier in this kernel)
    const int half = lane >> 5, l31 = lane & 31;
    int co_i = item % p.n_co_items;                                 // output channel item fastest: its waves share the pixels
    const int t = item / p.n_co_items;
    int px_t = t % p.n_px, b = t / p.n_px;
    // (the divisions run on the vector unit; the compiler knows the results are uniform and folds a readfirstlane away, yet
    //  keeps base pointers and buffer resources derived from them in vector registers -- a waterfall loop around every load)
    asm volatile("" : "+s"(co_i), "+s"(px_t), "+s"(b));
    constexpr int CB = 2;                                           // 32-channel blocks per wave
    const int px0 = 32 * NB * px_t, co0 = 64 * co_i;
    const int Cin = p.c0 + p.c1, HW = p.HW, n = p.nchunks;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0 + (size_t)b * p.c0 * HW), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in1 ? p.in1 + (size_t)b * p.c1 * HW : p.in0), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wb), 0, -1, 0x00020000);

    // the lane's pixel pair and channel group: channel k of the group is a scalar offset further on
    const unsigned x_voff = 4u * (unsigned)((4 * half) * HW + px0 + NB * l31);
    const unsigned a_lane16 = 16u * (unsigned)lane, a_lane8 = 8u * (unsigned)lane;

    float* const gnL = smem + wave_u * (2 * Cin);                   // [input channel][scale, shift] of this wave's image
    if constexpr (PRO != 0) {
        for (int i = lane; i < Cin; i += 64) {
            gnL[2 * i] = p.gn_scale[(size_t)b * Cin + i];
            gnL[2 * i + 1] = p.gn_shift[(size_t)b * Cin + i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // written and read by this wave only
        __builtin_amdgcn_wave_barrier();
    }

    pwb_f32x16 acc[CB][NB];                                         // [channel block][pixel parity]
#pragma unroll
    for (int x = 0; x < CB; ++x)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][nb][r] = 0.0f;

    struct XRegs {
        pwb_f2 x[4];             // channels 4 g .. 4 g + 3 of the chunk, the lane's pixel pair (NB == 1: .x only)
    };
    auto ld = [&](const __amdgpu_buffer_rsrc_t rs, unsigned soff) {
        if constexpr (NB == 2) return __builtin_bit_cast(pwb_f2, __builtin_amdgcn_raw_buffer_load_b64(rs, x_voff, soff, 0));
        else return pwb_f2{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, x_voff, soff, 0)), 0.0f};
    };
    pwb_u4 ua[CB];               // (U_hi, U_mid) of the current chunk
    pwb_u2 ul[CB];               // U_lo
    pwb_u4 ub[CB];               // 32-pixel form only: the filters of the chunk after (a second set: requested a whole chunk
    pwb_u2 um[CB];               // before their use -- with one set the L2 latency of every chunk's filters was in the open)
    // activations: one chunk ahead in registers (two sets); filters: requested as soon as the previous chunk's MFMAs are
    // issued -- the SIMD's other waves cover that latency, there is no barrier to hold them back
    auto load_x = [&](int c, XRegs& r) {
#ifdef PWB_DIAG_NO_X_LOADS
        if (c > 1) return;          // diagnostic: results wrong
#endif
        const int cc0 = 8 * c;
        if (cc0 < p.c0) {                                           // the concat seam lies on a chunk boundary (launcher); a
            const unsigned soff = 4u * (unsigned)(cc0 * HW);        // uniform branch, not a select of the resource (waterfall)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                r.x[k] = ld(rs0, soff + 4u * (unsigned)(k * HW));
        } else {
            const unsigned soff = 4u * (unsigned)((cc0 - p.c0) * HW);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                r.x[k] = ld(rs1, soff + 4u * (unsigned)(k * HW));
        }
    };
    auto load_a = [&](int c) {
#ifdef PWB_DIAG_NO_FILTER_LOADS
        if (c > 1) return;          // diagnostic: results wrong
#endif
        const unsigned s0 = 3072u * (unsigned)(c * p.n_co64 + co_i);
#pragma unroll
        for (int x = 0; x < CB; ++x) {
            ua[x] = __builtin_amdgcn_raw_buffer_load_b128(rsw, a_lane16, s0 + 1024u * (unsigned)x, 0);
            ul[x] = __builtin_amdgcn_raw_buffer_load_b64(rsw, a_lane8, s0 + 2048u + 512u * (unsigned)x, 0);
        }
    };
    auto load_b = [&](int c) {
#ifdef PWB_DIAG_NO_FILTER_LOADS
        if (c > 1) return;
#endif
        const unsigned s0 = 3072u * (unsigned)(c * p.n_co64 + co_i);
#pragma unroll
        for (int x = 0; x < CB; ++x) {
            ub[x] = __builtin_amdgcn_raw_buffer_load_b128(rsw, a_lane16, s0 + 1024u * (unsigned)x, 0);
            um[x] = __builtin_amdgcn_raw_buffer_load_b64(rsw, a_lane8, s0 + 2048u + 512u * (unsigned)x, 0);
        }
    };
    struct Tup { pwb_u4 hm, mh, lh; };                           // the three B operands of a 32-pixel block
    // GroupNorm (+ SiLU) and the exact split of the lane's four channels of pixel block nb: truncations and exact differences
    // (plain vector instructions on purpose: packed-f32 ones are not hidden by the bf16 MFMA, tools/bf16_issue_probe.hip)
    auto make = [&](int c, const XRegs& r, int nb) {
        pwb_f4 g01 = {1.0f, 0.0f, 1.0f, 0.0f}, g23 = g01;          // (scale, shift) of channels 0, 1 and 2, 3 of the lane's group
        if constexpr (PRO != 0) {
            const float* gp = gnL + 2 * (8 * c + 4 * half);
            g01 = *reinterpret_cast<const pwb_f4*>(gp);
            g23 = *reinterpret_cast<const pwb_f4*>(gp + 4);
        }
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float e = nb ? r.x[k].y : r.x[k].x;
            if constexpr (PRO != 0) {
                const float sc = k == 0 ? g01.x : (k == 1 ? g01.z : (k == 2 ? g23.x : g23.z));
                const float sh = k == 0 ? g01.y : (k == 1 ? g01.w : (k == 2 ? g23.y : g23.w));
                e = e * sc + sh;
                if constexpr (PRO == 2) e = pwb_silu(e);
            }
            v[k] = e;
        }
        unsigned hi[2], mid[2], lo[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float v0 = v[2 * q], v1 = v[2 * q + 1];
            const unsigned h0 = __float_as_uint(v0) & 0xffff0000u, h1 = __float_as_uint(v1) & 0xffff0000u;
            const float r0 = v0 - __uint_as_float(h0), r1 = v1 - __uint_as_float(h1);
            const unsigned m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
            const float l0 = r0 - __uint_as_float(m0), l1 = r1 - __uint_as_float(m1);
            hi[q] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
            mid[q] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
            lo[q] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
        }
        return Tup{pwb_u4{hi[0], hi[1], mid[0], mid[1]}, pwb_u4{mid[0], mid[1], hi[0], hi[1]}, pwb_u4{lo[0], lo[1], hi[0], hi[1]}};
    };
    auto mm_with = [&](const Tup& t, auto nb_tag, const pwb_u4 (&fa)[CB], const pwb_u2 (&fl)[CB]) {
        constexpr int nb = decltype(nb_tag)::value;
#pragma unroll
        for (int x = 0; x < CB; ++x) {
            const pwb_u4 a_hl = {fa[x].x, fa[x].y, fl[x].x, fl[x].y};
            acc[x][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pwb_bf16x8, fa[x]), __builtin_bit_cast(pwb_bf16x8, t.hm), acc[x][nb], 0, 0, 0);
            acc[x][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pwb_bf16x8, fa[x]), __builtin_bit_cast(pwb_bf16x8, t.mh), acc[x][nb], 0, 0, 0);
            acc[x][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pwb_bf16x8, a_hl), __builtin_bit_cast(pwb_bf16x8, t.lh), acc[x][nb], 0, 0, 0);
        }
    };
    auto mm = [&](const Tup& t, auto nb_tag) { mm_with(t, nb_tag, ua, ul); };
    using Z = std::integral_constant<int, 0>;
    using O = std::integral_constant<int, 1>;

    XRegs ra, rb;
    load_x(0, ra);
    load_a(0);
    if constexpr (NB == 2) {
        auto compute = [&](int c, const XRegs& r) {
            mm(make(c, r, 0), Z{});
            mm(make(c, r, 1), O{});
        };
        int c = 0;
        for (; c + 1 < n; c += 2) {
            load_x(c + 1, rb);
            compute(c, ra);
            load_a(c + 1);
            if (c + 2 < n) load_x(c + 2, ra);
            compute(c + 1, rb);
            if (c + 2 < n) load_a(c + 2);
        }
        if (c < n) compute(c, ra);
    } else {
        // one pixel block per wave leaves registers for a software pipeline: the operands of chunk c + 1 are put together
        // BETWEEN the MFMAs of chunk c (a wave issues in order: vector work behind six MFMAs waits for all of them)
        auto weave = [&]() {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);       // eight vector instructions
            }
        };
        if (n > 1) { load_x(1, rb); load_b(1); }
        Tup t = make(0, ra, 0);
        int c = 0;
        for (; c + 2 < n; c += 2) {             // t: operands of chunk c (even); filters: even chunks in (ua, ul), odd in (ub, um)
            load_x(c + 2, ra);
            Tup u = make(c + 1, rb, 0);
            mm_with(t, Z{}, ua, ul);
            weave();
            load_a(c + 2);
            if (c + 3 < n) load_x(c + 3, rb);
            t = make(c + 2, ra, 0);
            mm_with(u, Z{}, ub, um);
            weave();
            if (c + 3 < n) load_b(c + 3);
        }
        if (c + 1 < n) {
            Tup u = make(c + 1, rb, 0);
            mm_with(t, Z{}, ua, ul);
            mm_with(u, Z{}, ub, um);
        } else {
            mm_with(t, Z{}, ua, ul);
        }
    }

    // ---- epilogue: bias + per-sample channel bias + residual, NCHW stores of the pixel pair; GroupNorm partials (one slot
    // per 32 consecutive pixels)
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)b * p.Cout * HW, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.residual ? p.residual + (size_t)b * p.Cout * HW : p.out), 0, -1, 0x00020000);
    const unsigned o_voff = 4u * (unsigned)((4 * half) * HW + px0 + NB * l31);
    const int slots = HW / 32;
#pragma unroll
    for (int x = 0; x < CB; ++x) {
        // (all residual / bias operands of the block are requested before the first is used)
        float add[16];
        pwb_f2 res[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cs = co0 + 32 * x + 8 * (r >> 2) + (r & 3);          // + 4 half: the lane's part
            const int col = cs + 4 * half;
            add[r] = 0.0f;
            if (p.bias) add[r] += p.bias[col];
            if (p.chan_bias) add[r] += p.chan_bias[(size_t)b * p.chan_bias_stride + col];
            res[r] = pwb_f2{0.0f, 0.0f};
            if (p.residual) {
                if constexpr (NB == 2) res[r] = __builtin_bit_cast(pwb_f2, __builtin_amdgcn_raw_buffer_load_b64(rsr, o_voff, 4u * (unsigned)(cs * HW), 0));
                else res[r].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsr, o_voff, 4u * (unsigned)(cs * HW), 0));
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cs = co0 + 32 * x + 8 * (r >> 2) + (r & 3);
            float vv[2] = {0.0f, 0.0f};
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                vv[nb] = acc[x][nb][r] + add[r] + (nb ? res[r].y : res[r].x);
                if (p.relu) vv[nb] = fmaxf(vv[nb], 0.0f);
            }
            if constexpr (NB == 2) __builtin_amdgcn_raw_buffer_store_b64(pwb_u2{__float_as_uint(vv[0]), __float_as_uint(vv[1])}, rso, o_voff, 4u * (unsigned)(cs * HW), 0);
            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vv[0]), rso, o_voff, 4u * (unsigned)(cs * HW), 0);
            if (p.stats) {
                // a slot = 32 CONSECUTIVE pixels, summed over the same binary tree (pairs, fours, ... of neighbours) in both
                // forms, so that the partials -- and the GroupNorm of the next layer -- do not depend on the form, i.e. on the
                // batch: one pixel per lane: five lane levels; a pixel pair per lane: the pair, then four lane levels
                const int co = cs + 4 * half;
                if constexpr (NB == 1) {
                    const float s1 = pwb_half_wave_sum(vv[0]);
                    const float d = vv[0] - s1 * (1.0f / 32.0f);
                    float dd;                  // the ROUNDED square (the compiler would fuse it into the first add of the tree, in
                    asm volatile("v_mul_f32 %0, %1, %1" : "=v"(dd) : "v"(d));       // one lane of each pair only)
                    const float q = pwb_half_wave_sum(dd);
                    if (l31 == 0)
                        reinterpret_cast<float4*>(p.stats)[((size_t)b * p.Cout + co) * slots + px_t] = make_float4(32.0f, s1, q, 0.0f);
                } else {
                    const float s1 = pwb_row16_sum(vv[0] + vv[1]);
                    const float mean = s1 * (1.0f / 32.0f);
                    const float d0 = vv[0] - mean, d1 = vv[1] - mean;
                    float q0, q1;              // two ROUNDED squares as in the other form (the compiler would fuse one into an fma)
                    asm volatile("v_mul_f32 %0, %2, %2\n\tv_mul_f32 %1, %3, %3" : "=&v"(q0), "=&v"(q1) : "v"(d0), "v"(d1));
                    const float q = pwb_row16_sum(q0 + q1);
                    if ((l31 & 15) == 0)
                        reinterpret_cast<float4*>(p.stats)[((size_t)b * p.Cout + co) * slots + 2 * px_t + (l31 >> 4)] = make_float4(32.0f, s1, q, 0.0f);
                }
            }
        }
    }
}

// ---- K-SPLIT form (round 4; tile_cfg 35): the small levels' 1x1 layers (256 -> 256 at 16x16: 2048 wave items of 32 pixels x 64
// channels at batch 64, two per SIMD) are LATENCY-bound -- a wave is parked at s_waitcnt 60 % of its life (profiles/r04/
// sol_pointwise.txt), every one of its 32 chunks waits for its own loads.  Here the four waves of a workgroup share ONE item and
// split its input channels four ways (a quarter of the chain each, all in flight together), put their partial accumulators into
// LDS, and each wave sums -- in the fixed order ((k0 + k1) + k2) + k3 -- and stores a quarter of the item's 64 channels.
// Another summation order than the forms above: other bits, so the choice between them is by layer SHAPE only (launcher).
#ifndef PWBK_VAR
#define PWBK_VAR 0
#endif
template <int PRO>
__global__ void __launch_bounds__(64 * PWB_WAVES, 4) conv_pwbk_kernel(const PwbParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int item;
    {   // XCD-aware bijective remap (conv_mfma.hip)
        const int L = blockIdx.x, nwg = p.nwg;
        const int xcd = L & 7, slot = L >> 3, q = nwg >> 3, r = nwg & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int co_i = item % p.n_co_items;
    const int t = item / p.n_co_items;
    int px_t = t % p.n_px, b = t / p.n_px;
    asm volatile("" : "+s"(co_i), "+s"(px_t), "+s"(b));
    constexpr int CB = 2;
    const int px0 = 32 * px_t, co0 = 64 * co_i;
    const int Cin = p.c0 + p.c1, HW = p.HW;
    const int nl = p.nchunks / PWB_WAVES, cb = wave_u * nl;         // this wave's chunks: cb .. cb + nl - 1

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0 + (size_t)b * p.c0 * HW), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in1 ? p.in1 + (size_t)b * p.c1 * HW : p.in0), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wb), 0, -1, 0x00020000);
    const unsigned x_voff = 4u * (unsigned)((4 * half) * HW + px0 + l31);
    const unsigned a_lane16 = 16u * (unsigned)lane, a_lane8 = 8u * (unsigned)lane;

    // LDS: [4 waves][32 accumulator registers][64 lanes] partials, then a (scale, shift) table of the wave's own channels each
    float* const P_lds = smem;
    float* const gnL = smem + PWB_WAVES * 32 * 64 + wave_u * (2 * 8 * nl);
    if constexpr (PRO != 0) {
        for (int i = lane; i < 8 * nl; i += 64) {
            gnL[2 * i] = p.gn_scale[(size_t)b * Cin + 8 * cb + i];
            gnL[2 * i + 1] = p.gn_shift[(size_t)b * Cin + 8 * cb + i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // written and read by this wave only
        __builtin_amdgcn_wave_barrier();
    }

    pwb_f32x16 acc[CB];
#pragma unroll
    for (int x = 0; x < CB; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;
    struct XRegs { float x[4]; };
    auto load_x = [&](int c, XRegs& r) {                            // c: absolute chunk
        const int cc0 = 8 * c;
        if (cc0 < p.c0) {
            const unsigned soff = 4u * (unsigned)(cc0 * HW);
#pragma unroll
            for (int k = 0; k < 4; ++k) r.x[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs0, x_voff, soff + 4u * (unsigned)(k * HW), 0));
        } else {
            const unsigned soff = 4u * (unsigned)((cc0 - p.c0) * HW);
#pragma unroll
            for (int k = 0; k < 4; ++k) r.x[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, x_voff, soff + 4u * (unsigned)(k * HW), 0));
        }
    };
    auto load_f = [&](int c, pwb_u4 (&fa)[CB], pwb_u2 (&fl)[CB]) {
        const unsigned s0 = 3072u * (unsigned)(c * p.n_co64 + co_i);
#pragma unroll
        for (int x = 0; x < CB; ++x) {
            fa[x] = __builtin_amdgcn_raw_buffer_load_b128(rsw, a_lane16, s0 + 1024u * (unsigned)x, 0);
#if (PWBK_VAR & 1) != 0           // timing-only builds (wrong results): bit 0 no U_lo fetch, bit 1 no filter fetch at all
            fl[x] = pwb_u2{fa[x].z, fa[x].w};
#else
            fl[x] = __builtin_amdgcn_raw_buffer_load_b64(rsw, a_lane8, s0 + 2048u + 512u * (unsigned)x, 0);
#endif
#if (PWBK_VAR & 2) != 0
            fa[x] = pwb_u4{a_lane16 + (unsigned)c, a_lane8, s0, a_lane16 ^ s0};
            fl[x] = pwb_u2{fa[x].z, fa[x].w};
#endif
        }
    };
    struct Tup { pwb_u4 hm, mh, lh; };
    auto make = [&](int cl, const XRegs& r) {                       // cl: the wave's local chunk (its GroupNorm table's index)
        pwb_f4 g01 = {1.0f, 0.0f, 1.0f, 0.0f}, g23 = g01;
        if constexpr (PRO != 0) {
            const float* gp = gnL + 2 * (8 * cl + 4 * half);
            g01 = *reinterpret_cast<const pwb_f4*>(gp);
            g23 = *reinterpret_cast<const pwb_f4*>(gp + 4);
        }
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float e = r.x[k];
            if constexpr (PRO != 0) {
                const float sc = k == 0 ? g01.x : (k == 1 ? g01.z : (k == 2 ? g23.x : g23.z));
                const float sh = k == 0 ? g01.y : (k == 1 ? g01.w : (k == 2 ? g23.y : g23.w));
                e = e * sc + sh;
                if constexpr (PRO == 2) e = pwb_silu(e);
            }
            v[k] = e;
        }
        unsigned hi[2], mid[2], lo[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float v0 = v[2 * q], v1 = v[2 * q + 1];
            const unsigned h0 = __float_as_uint(v0) & 0xffff0000u, h1 = __float_as_uint(v1) & 0xffff0000u;
            const float r0 = v0 - __uint_as_float(h0), r1 = v1 - __uint_as_float(h1);
            const unsigned m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
            const float l0 = r0 - __uint_as_float(m0), l1 = r1 - __uint_as_float(m1);
            hi[q] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
            mid[q] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
            lo[q] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
        }
        return Tup{pwb_u4{hi[0], hi[1], mid[0], mid[1]}, pwb_u4{mid[0], mid[1], hi[0], hi[1]}, pwb_u4{lo[0], lo[1], hi[0], hi[1]}};
    };
    auto mm = [&](const Tup& t, const pwb_u4 (&fa)[CB], const pwb_u2 (&fl)[CB]) {
#pragma unroll
        for (int x = 0; x < CB; ++x) {
            const pwb_u4 a_hl = {fa[x].x, fa[x].y, fl[x].x, fl[x].y};
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pwb_bf16x8, fa[x]), __builtin_bit_cast(pwb_bf16x8, t.hm), acc[x], 0, 0, 0);
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pwb_bf16x8, fa[x]), __builtin_bit_cast(pwb_bf16x8, t.mh), acc[x], 0, 0, 0);
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pwb_bf16x8, a_hl), __builtin_bit_cast(pwb_bf16x8, t.lh), acc[x], 0, 0, 0);
        }
    };
    // every operand of the wave's (short) chain is requested up front where the registers allow: activations four chunks deep,
    // filters two
    constexpr int XD = 4;
    XRegs xr[XD];
    pwb_u4 ua[CB], ub[CB];
    pwb_u2 ul[CB], um[CB];
#pragma unroll
    for (int i = 0; i < XD; ++i)
        if (i < nl) load_x(cb + i, xr[i]);
    load_f(cb, ua, ul);
    if (nl > 1) load_f(cb + 1, ub, um);
    for (int c = 0; c < nl; c += XD) {
#pragma unroll
        for (int i = 0; i < XD; ++i) {
            if (c + i < nl) {
                const Tup t = make(c + i, xr[i]);
                if (c + i + XD < nl) load_x(cb + c + i + XD, xr[i]);
                if ((i & 1) == 0) {
                    mm(t, ua, ul);
                    if (c + i + 2 < nl) load_f(cb + c + i + 2, ua, ul);
                } else {
                    mm(t, ub, um);
                    if (c + i + 2 < nl) load_f(cb + c + i + 2, ub, um);
                }
            }
        }
    }

    // ---- the four partial tiles through LDS; wave w sums and stores accumulator rows 8 (w & 1) .. + 7 of channel block w >> 1
#pragma unroll
    for (int x = 0; x < CB; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) P_lds[((wave_u * 2 + x) * 16 + r) * 64 + lane] = acc[x][r];
    __syncthreads();
    const int x = wave_u >> 1, rb = 8 * (wave_u & 1);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)b * p.Cout * HW, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.residual ? p.residual + (size_t)b * p.Cout * HW : p.out), 0, -1, 0x00020000);
    const unsigned o_voff = 4u * (unsigned)((4 * half) * HW + px0 + l31);
    const int slots = HW / 32;
    float add[8], res[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int r = rb + rr;
        const int cs = co0 + 32 * x + 8 * (r >> 2) + (r & 3);
        const int col = cs + 4 * half;
        add[rr] = 0.0f;
        if (p.bias) add[rr] += p.bias[col];
        if (p.chan_bias) add[rr] += p.chan_bias[(size_t)b * p.chan_bias_stride + col];
        res[rr] = 0.0f;
        if (p.residual) res[rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsr, o_voff, 4u * (unsigned)(cs * HW), 0));
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int r = rb + rr;
        const int cs = co0 + 32 * x + 8 * (r >> 2) + (r & 3);
        float part[PWB_WAVES];
#pragma unroll
        for (int k = 0; k < PWB_WAVES; ++k) part[k] = P_lds[((k * 2 + x) * 16 + r) * 64 + lane];
        float v = ((part[0] + part[1]) + part[2]) + part[3];
        v = v + add[rr] + res[rr];
        if (p.relu) v = fmaxf(v, 0.0f);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rso, o_voff, 4u * (unsigned)(cs * HW), 0);
        if (p.stats) {          // (the 32-pixel form's slot and summation tree, conv_pwb_kernel<PRO, 1>)
            const int co = cs + 4 * half;
            const float s1 = pwb_half_wave_sum(v);
            const float d = v - s1 * (1.0f / 32.0f);
            float dd;
            asm volatile("v_mul_f32 %0, %1, %1" : "=v"(dd) : "v"(d));
            const float q = pwb_half_wave_sum(dd);
            if (l31 == 0)
                reinterpret_cast<float4*>(p.stats)[((size_t)b * p.Cout + co) * slots + px_t] = make_float4(32.0f, s1, q, 0.0f);
        }
    }
}

// ---- STAGED form (round 4; tile_cfg 34): for a layer with many output channels (q, k, v of an attention block: 256 -> 768)
// the kernel above splits every activation once per 64-channel item that reads it -- twelve times -- and the split (GroupNorm,
// three terms, packing: ~15 vector instructions per element) is what its loop consists of.  Here a workgroup owns 64 pixels of
// one image and ALL output channels: its waves (one per 64-channel item) first put the pixels' B operands -- GroupNorm applied,
// split, packed exactly as above -- into LDS ONCE (Cin x 64 pixels x 6 bytes: 96 KB at 256 channels), then every wave runs the
// contraction of its own channel item with operands that are two LDS reads per chunk and pixel block; filters from global memory
// two chunks ahead as before.  The chain of MFMAs of an output is the one of the forms above: same bits.
constexpr int PWBS_MAX_WAVES = 12;
template <int PRO>
__global__ void __launch_bounds__(64 * PWBS_MAX_WAVES) conv_pwbs_kernel(const PwbParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int wg;
    {   // XCD-aware bijective remap (conv_mfma.hip)
        const int L = blockIdx.x, nwg = p.nwg;
        const int xcd = L & 7, slot = L >> 3, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = p.n_co_items;                                // one wave per 64-channel item
    const int half = lane >> 5, l31 = lane & 31;
    int px_t = wg % p.n_px, b = wg / p.n_px;
    asm volatile("" : "+s"(px_t), "+s"(b));
    constexpr int CB = 2, NB = 2;
    const int co_i = wave_u;
    const int px0 = 64 * px_t, co0 = 64 * co_i;
    const int Cin = p.c0 + p.c1, HW = p.HW, n = p.nchunks;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0 + (size_t)b * p.c0 * HW), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in1 ? p.in1 + (size_t)b * p.c1 * HW : p.in0), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wb), 0, -1, 0x00020000);
    const unsigned x_voff = 4u * (unsigned)((4 * half) * HW + px0 + NB * l31);
    const unsigned a_lane16 = 16u * (unsigned)lane, a_lane8 = 8u * (unsigned)lane;

    // LDS: [chunk][pixel block][64 lanes][(hi, mid) 4 dwords] then [chunk][pixel block][64 lanes][lo 2 dwords], then the GroupNorm table
    unsigned* const S_hm = reinterpret_cast<unsigned*>(smem);
    unsigned* const S_lo = S_hm + (size_t)n * NB * 64 * 4;
    float* const gnL = reinterpret_cast<float*>(S_lo + (size_t)n * NB * 64 * 2);      // [input channel][scale, shift]
    if constexpr (PRO != 0) {
        for (int i = tid; i < Cin; i += 64 * nwaves) {
            gnL[2 * i] = p.gn_scale[(size_t)b * Cin + i];
            gnL[2 * i + 1] = p.gn_shift[(size_t)b * Cin + i];
        }
        __syncthreads();
    }
    // ---- stage: wave w splits chunks w, w + nwaves, ... (the lane mapping, GroupNorm and split of conv_pwb_kernel's `make`)
    for (int c = wave_u; c < n; c += nwaves) {
        pwb_f2 xr[4];
        const int cc0 = 8 * c;
        if (cc0 < p.c0) {
            const unsigned soff = 4u * (unsigned)(cc0 * HW);
#pragma unroll
            for (int k = 0; k < 4; ++k) xr[k] = __builtin_bit_cast(pwb_f2, __builtin_amdgcn_raw_buffer_load_b64(rs0, x_voff, soff + 4u * (unsigned)(k * HW), 0));
        } else {
            const unsigned soff = 4u * (unsigned)((cc0 - p.c0) * HW);
#pragma unroll
            for (int k = 0; k < 4; ++k) xr[k] = __builtin_bit_cast(pwb_f2, __builtin_amdgcn_raw_buffer_load_b64(rs1, x_voff, soff + 4u * (unsigned)(k * HW), 0));
        }
        pwb_f4 g01 = {1.0f, 0.0f, 1.0f, 0.0f}, g23 = g01;
        if constexpr (PRO != 0) {
            const float* gp = gnL + 2 * (8 * c + 4 * half);
            g01 = *reinterpret_cast<const pwb_f4*>(gp);
            g23 = *reinterpret_cast<const pwb_f4*>(gp + 4);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float e = nb ? xr[k].y : xr[k].x;
                if constexpr (PRO != 0) {
                    const float sc = k == 0 ? g01.x : (k == 1 ? g01.z : (k == 2 ? g23.x : g23.z));
                    const float sh = k == 0 ? g01.y : (k == 1 ? g01.w : (k == 2 ? g23.y : g23.w));
                    e = e * sc + sh;
                    if constexpr (PRO == 2) e = pwb_silu(e);
                }
                v[k] = e;
            }
            unsigned hi[2], mid[2], lo[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float v0 = v[2 * q], v1 = v[2 * q + 1];
                const unsigned h0 = __float_as_uint(v0) & 0xffff0000u, h1 = __float_as_uint(v1) & 0xffff0000u;
                const float r0 = v0 - __uint_as_float(h0), r1 = v1 - __uint_as_float(h1);
                const unsigned m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
                const float l0 = r0 - __uint_as_float(m0), l1 = r1 - __uint_as_float(m1);
                hi[q] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
                mid[q] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
                lo[q] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
            }
            *reinterpret_cast<pwb_u4*>(S_hm + ((size_t)(c * NB + nb) * 64 + lane) * 4) = pwb_u4{hi[0], hi[1], mid[0], mid[1]};
            *reinterpret_cast<pwb_u2*>(S_lo + ((size_t)(c * NB + nb) * 64 + lane) * 2) = pwb_u2{lo[0], lo[1]};
        }
    }
    __syncthreads();

    // ---- contract: this wave's 64 channels x the 64 pixels, filters two chunks ahead in registers
    pwb_f32x16 acc[CB][NB];
#pragma unroll
    for (int x = 0; x < CB; ++x)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][nb][r] = 0.0f;
    pwb_u4 ua[CB], ub[CB];
    pwb_u2 ul[CB], um[CB];
    auto load_f = [&](int c, pwb_u4 (&fa)[CB], pwb_u2 (&fl)[CB]) {
        const unsigned s0 = 3072u * (unsigned)(c * p.n_co64 + co_i);
#pragma unroll
        for (int x = 0; x < CB; ++x) {
            fa[x] = __builtin_amdgcn_raw_buffer_load_b128(rsw, a_lane16, s0 + 1024u * (unsigned)x, 0);
            fl[x] = __builtin_amdgcn_raw_buffer_load_b64(rsw, a_lane8, s0 + 2048u + 512u * (unsigned)x, 0);
        }
    };
    auto chunk = [&](int c, const pwb_u4 (&fa)[CB], const pwb_u2 (&fl)[CB]) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const pwb_u4 hm = *reinterpret_cast<const pwb_u4*>(S_hm + ((size_t)(c * NB + nb) * 64 + lane) * 4);
            const pwb_u2 lo = *reinterpret_cast<const pwb_u2*>(S_lo + ((size_t)(c * NB + nb) * 64 + lane) * 2);
            const pwb_u4 mh = {hm.z, hm.w, hm.x, hm.y}, lh = {lo.x, lo.y, hm.x, hm.y};
#pragma unroll
            for (int x = 0; x < CB; ++x) {
                const pwb_u4 a_hl = {fa[x].x, fa[x].y, fl[x].x, fl[x].y};
                acc[x][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pwb_bf16x8, fa[x]), __builtin_bit_cast(pwb_bf16x8, hm), acc[x][nb], 0, 0, 0);
                acc[x][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pwb_bf16x8, fa[x]), __builtin_bit_cast(pwb_bf16x8, mh), acc[x][nb], 0, 0, 0);
                acc[x][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pwb_bf16x8, a_hl), __builtin_bit_cast(pwb_bf16x8, lh), acc[x][nb], 0, 0, 0);
            }
        }
    };
    load_f(0, ua, ul);
    if (n > 1) load_f(1, ub, um);
    {
        int c = 0;
        for (; c + 1 < n; c += 2) {
            chunk(c, ua, ul);
            if (c + 2 < n) load_f(c + 2, ua, ul);
            chunk(c + 1, ub, um);
            if (c + 3 < n) load_f(c + 3, ub, um);
        }
        if (c < n) chunk(c, ua, ul);
    }

    // ---- epilogue: as conv_pwb_kernel<PRO, 2>
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)b * p.Cout * HW, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.residual ? p.residual + (size_t)b * p.Cout * HW : p.out), 0, -1, 0x00020000);
    const unsigned o_voff = 4u * (unsigned)((4 * half) * HW + px0 + NB * l31);
    const int slots = HW / 32;
#pragma unroll
    for (int x = 0; x < CB; ++x) {
        float add[16];
        pwb_f2 res[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cs = co0 + 32 * x + 8 * (r >> 2) + (r & 3);
            const int col = cs + 4 * half;
            add[r] = 0.0f;
            if (p.bias) add[r] += p.bias[col];
            if (p.chan_bias) add[r] += p.chan_bias[(size_t)b * p.chan_bias_stride + col];
            res[r] = pwb_f2{0.0f, 0.0f};
            if (p.residual) res[r] = __builtin_bit_cast(pwb_f2, __builtin_amdgcn_raw_buffer_load_b64(rsr, o_voff, 4u * (unsigned)(cs * HW), 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cs = co0 + 32 * x + 8 * (r >> 2) + (r & 3);
            float vv[2];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                vv[nb] = acc[x][nb][r] + add[r] + (nb ? res[r].y : res[r].x);
                if (p.relu) vv[nb] = fmaxf(vv[nb], 0.0f);
            }
            __builtin_amdgcn_raw_buffer_store_b64(pwb_u2{__float_as_uint(vv[0]), __float_as_uint(vv[1])}, rso, o_voff, 4u * (unsigned)(cs * HW), 0);
            if (p.stats) {
                const int co = cs + 4 * half;
                const float s1 = pwb_row16_sum(vv[0] + vv[1]);
                const float mean = s1 * (1.0f / 32.0f);
                const float d0 = vv[0] - mean, d1 = vv[1] - mean;
                float q0, q1;
                asm volatile("v_mul_f32 %0, %2, %2\n\tv_mul_f32 %1, %3, %3" : "=&v"(q0), "=&v"(q1) : "v"(d0), "v"(d1));
                const float q = pwb_row16_sum(q0 + q1);
                if ((l31 & 15) == 0)
                    reinterpret_cast<float4*>(p.stats)[((size_t)b * p.Cout + co) * slots + 2 * px_t + (l31 >> 4)] = make_float4(32.0f, s1, q, 0.0f);
            }
        }
    }
}

static bool pwbk_on() { static const bool on = [] { const char* e = std::getenv("SISIC_POINTWISE_KSPLIT"); return !e || std::atoi(e) != 0; }(); return on; }
template <int PRO>
static int launch_pwbk(sisic_ctx* ctx, PwbParams& p, int Cin, hipStream_t s) {
    p.n_co_items = p.Cout / 64;
    p.n_px = p.HW / 32;
    const int64_t nitems = (int64_t)p.B * p.n_px * p.n_co_items;
    SISIC_REQUIRE(nitems > 0 && nitems < (int64_t(1) << 31), "conv2d(pointwise bf16x3, K-split): grid too large");
    p.nitems = (int)nitems;
    p.nwg = p.nitems;
    const size_t lds = sizeof(float) * (size_t)(PWB_WAVES * 32 * 64 + (PRO ? 2 * Cin : 0));
    static std::atomic<uint64_t> opt{0};
    auto kern = conv_pwbk_kernel<PRO>;
    SISIC_TRY(ensure_dynamic_lds(ctx, reinterpret_cast<const void*>(kern), (int)lds, opt));
    hipLaunchKernelGGL(kern, dim3(p.nwg), dim3(64 * PWB_WAVES), lds, s, p);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// staged form: LDS holds the image's 64 pixels of every input channel as split operands (6 bytes per value) and the GroupNorm table
static size_t pwbs_lds_bytes(int Cin) { return (size_t)(Cin / 8) * 2 * 64 * 24 + 8 * (size_t)Cin; }
static bool pwbs_applicable(const sisic_conv_args& a) {
    const int items = a.Cout / 64;
    return items >= 6 && items <= PWBS_MAX_WAVES && pwbs_lds_bytes(a.c0 + a.c1) <= 150 * 1024;
}
template <int PRO>
static int launch_pwbs(sisic_ctx* ctx, PwbParams& p, int Cin, hipStream_t s) {
    p.n_co_items = p.Cout / 64;
    p.n_px = p.HW / 64;
    const int64_t nwg = (int64_t)p.B * p.n_px;
    SISIC_REQUIRE(nwg > 0 && nwg < (int64_t(1) << 31), "conv2d(pointwise bf16x3, staged): grid too large");
    p.nwg = (int)nwg;
    p.nitems = p.nwg * p.n_co_items;
    const size_t lds = pwbs_lds_bytes(Cin);
    static std::atomic<uint64_t> opt{0};
    auto kern = conv_pwbs_kernel<PRO>;
    SISIC_TRY(ensure_dynamic_lds(ctx, reinterpret_cast<const void*>(kern), (int)lds, opt));
    hipLaunchKernelGGL(kern, dim3(p.nwg), dim3(64 * p.n_co_items), lds, s, p);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// The shapes this kernel takes: whole 64-pixel and 64-channel tiles, whole 8-channel chunks with the concat seam on a chunk
// boundary, 8-byte aligned pixel pairs, 32-bit byte offsets inside an image of either operand and inside the filter tensor.
bool conv_pointwise_bf3_applicable(const sisic_conv_args& a) {
    if (a.ksize != 1 || a.stride != 1 || a.upsample) return false;
    const int HW = a.Hin * a.Win, Cin = a.c0 + a.c1;
    if (HW % 64 != 0 || Cin % 8 != 0 || (a.c1 != 0 && a.c0 % 8 != 0) || a.Cout % 64 != 0) return false;
    if (((reinterpret_cast<uintptr_t>(a.in0) | reinterpret_cast<uintptr_t>(a.in1) | reinterpret_cast<uintptr_t>(a.out) |
          reinterpret_cast<uintptr_t>(a.residual)) & 7) != 0) return false;
    if (4.0 * std::max(std::max(a.c0, a.c1), a.Cout) * HW >= 4294967296.0 || 6.0 * conv_cin_pad(Cin, 1) * conv_cout_pad(a.Cout) >= 4294967296.0) return false;
    if (a.gn_scale && Cin > 1024) return false;               // GroupNorm operands of an image: a table per wave in LDS
    return true;
}

template <int PRO, int NB>
static int launch_pwb(sisic_ctx* ctx, PwbParams& p, int Cin, hipStream_t s) {
    p.n_co_items = p.Cout / 64;
    p.n_px = p.HW / (32 * NB);
    const int64_t nitems = (int64_t)p.B * p.n_px * p.n_co_items;
    SISIC_REQUIRE(nitems > 0 && nitems < (int64_t(1) << 31), "conv2d(pointwise bf16x3): grid too large");
    p.nitems = (int)nitems;
    p.nwg = (int)((nitems + PWB_WAVES - 1) / PWB_WAVES);
    const size_t lds = PRO ? sizeof(float) * (size_t)(PWB_WAVES * 2 * Cin) : 0;
    static std::atomic<uint64_t> opt{0};
    auto kern = conv_pwb_kernel<PRO, NB>;
    SISIC_TRY(ensure_dynamic_lds(ctx, reinterpret_cast<const void*>(kern), (int)lds, opt));
    hipLaunchKernelGGL(kern, dim3(p.nwg), dim3(64 * PWB_WAVES), lds, s, p);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

int launch_conv_pointwise_bf3(sisic_ctx* ctx, const sisic_conv_args& a, hipStream_t s) {
    SISIC_REQUIRE(conv_pointwise_bf3_applicable(a), "conv2d(pointwise bf16x3): shape not supported by tile_cfg 28");
    PwbParams p{};
    const int Cin = a.c0 + a.c1;
    p.in0 = a.in0; p.in1 = a.in1; p.c0 = a.c0; p.c1 = a.c1; p.B = a.B; p.HW = a.Hin * a.Win;
    p.wb = a.w_packed + 2 * (size_t)conv_cin_pad(Cin, 1) * conv_cout_pad(a.Cout);      // the third layout (pack_device.h)
    p.n_co64 = conv_cout_pad(a.Cout) / 64;
    p.bias = a.bias; p.Cout = a.Cout;
    p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift;
    p.chan_bias = a.chan_bias; p.chan_bias_stride = a.chan_bias_stride; p.residual = a.residual; p.relu = a.relu;
    p.out = a.out; p.stats = a.stats_out;
    p.nchunks = Cin / 8;
    const int pro = a.gn_scale == nullptr ? 0 : (a.gn_silu ? 2 : 1);
    // 64-pixel items where they give every SIMD at least two waves (1024 SIMDs), 32-pixel items otherwise: a lone wave has
    // nobody to hide its latencies.  (The choice depends on the batch; the bits of an output do not: its chain of MFMAs is the same.)
    // (tile_cfg 29 / 30 force the 32- / 64-pixel form: tests of their bit-equality)
    // the K-split form (tile_cfg 35 forces it) for the 16x16 and 8x8 levels' layers of up to 256 output channels -- by SHAPE only:
    // its bits are its own.  Measured at batch 64 (profiles/r04/conv_bench_pointwise_forms.txt): 256 -> 256 @16 27.3 vs 34.2 us,
    // 512 -> 256 @8 13.5 vs 24.9, 512 -> 256 @16 38.0 vs 38.7; NOT for 256 -> 768 (76.8 vs 50.7: twelve channel items already
    // fill the chip, and four waves per item fetch the item's filters in four strands)
    {
        const bool ks_ok = p.nchunks % (4 * PWB_WAVES) == 0;        // (whole groups of four chunks per wave; the concat seam lies on a chunk boundary)
        if (a.tile_cfg == 35) SISIC_REQUIRE(ks_ok, "conv2d(pointwise bf16x3, K-split): tile_cfg 35 needs a multiple of 128 input channels");
        if (a.tile_cfg == 35 || (a.tile_cfg == 0 && ks_ok && p.HW <= 256 && a.Cout <= 256 && pwbk_on())) {
            if (pro == 2) return launch_pwbk<2>(ctx, p, Cin, s);
            if (pro == 1) return launch_pwbk<1>(ctx, p, Cin, s);
            return launch_pwbk<0>(ctx, p, Cin, s);
        }
    }
    // the staged form (tile_cfg 34 forces it) where a layer has 6 .. 12 channel items and enough 64-pixel workgroups for the chip
    // (the choice depends on the batch; the bits do not)
    if (a.tile_cfg == 34) SISIC_REQUIRE(pwbs_applicable(a), "conv2d(pointwise bf16x3, staged): tile_cfg 34 needs 384 .. 768 output channels and at most %d input channels", (150 * 1024) / 392);
    if (a.tile_cfg == 34 || (a.tile_cfg == 0 && pwbs_applicable(a) && (int64_t)a.B * (p.HW / 64) >= 128)) {
        if (pro == 2) return launch_pwbs<2>(ctx, p, Cin, s);
        if (pro == 1) return launch_pwbs<1>(ctx, p, Cin, s);
        return launch_pwbs<0>(ctx, p, Cin, s);
    }
    const bool wide = a.tile_cfg == 29 ? false : (a.tile_cfg == 30 ? true : (int64_t)a.B * (p.HW / 64) * (a.Cout / 64) >= 2048);
    if (wide) {
        if (pro == 2) return launch_pwb<2, 2>(ctx, p, Cin, s);
        if (pro == 1) return launch_pwb<1, 2>(ctx, p, Cin, s);
        return launch_pwb<0, 2>(ctx, p, Cin, s);
    }
    if (pro == 2) return launch_pwb<2, 1>(ctx, p, Cin, s);
    if (pro == 1) return launch_pwb<1, 1>(ctx, p, Cin, s);
    return launch_pwb<0, 1>(ctx, p, Cin, s);
}

}  // namespace sisic
