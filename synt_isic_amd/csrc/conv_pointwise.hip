// conv_pointwise.hip -- 1x1 stride-1 convolutions (14 conv_shortcut, 6 fused q/k/v and 6 to_out projections per UNet forward)
// as a GEMM on the f32 MFMA pipe with as few other instructions as the operands allow.
//
// The generic implicit-GEMM kernel (conv_mfma.hip) spends 5.4 vector instructions per MFMA on these launches (ISA of its 1x1
// instances: float4 staging through scalars, per-element masks, 64-bit address arithmetic, one v_add per LDS operand
// read) and ran them at 0.51 of the f32 matrix peak for two rounds.  On gfx950 the f32 MFMA IS the SIMD's FMA array: every
// vector instruction displaces matrix work (tools/issue_probe.hip, profiles/r03/issue_probe.txt).  This kernel is the third
// Winograd form's recipe (conv_winograd_col.inc) applied to the plain GEMM:
//   D[co, px] = sum_ci W[co, ci] * act(X[ci, px])        workgroup = 64 output channels x 128 pixels of one image, 8 waves
//   * W never touches LDS: a second packing [8-channel chunk][32-channel block][lane][4 k-steps] (behind the generic packing
//     in the same buffer, written by the same pack functions) gives each wave its A fragments as one 16-byte buffer load per
//     8 input channels;
//   * X: 32-channel chunks, double-buffered in LDS as [channel][128 pixels]; a thread moves 2 x 4 consecutive pixels
//     (buffer_load_dwordx4 -> optional GroupNorm FMA / SiLU -> ds_write_b128); the chunk's plane offset rides the scalar
//     offset of the buffer load, the lane offsets are loop-invariant; GroupNorm operands of the image are staged in LDS once;
//   * B operands: ds_read_b32 at immediate offsets from one per-lane base (buffer parity is a compile-time constant);
//   * no masks: the launcher only takes shapes whose tiles are whole (H*W a multiple of 128, channels a multiple of 32,
//     the concat seam on a chunk boundary); everything else stays on conv_mfma.hip.
// Arithmetic: the same fp32 FMA chain in the same channel order as the generic kernel.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace sisic {

typedef float pw_f32x16 __attribute__((ext_vector_type(16)));
typedef float pw_v4f __attribute__((ext_vector_type(4)));

struct PwParams {
    const float* in0;
    const float* in1;
    int c0, c1, B, HW;
    const float* wpw;        // [Cin/8][cout_pad/32][64][4]
    int n_co32;
    const float* bias;
    int Cout;
    const float* gn_scale;
    const float* gn_shift;
    const float* chan_bias;
    int chan_bias_stride;
    const float* residual;
    int relu;
    float* out;
    float* stats;            // optional [B][Cout][n_px_tiles * 4][4]
    int n_px_tiles, n_co_tiles, nwg, nchunks;
};

constexpr int PW_PX = 128, PW_CIC = 32, PW_CO = 64, PW_XBUF = PW_CIC * PW_PX;

__device__ __forceinline__ float pw_half_wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
    return v + __shfl_xor(v, 16);
}
__device__ __forceinline__ float pw_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// NT = 32-pixel blocks per wave: 1 -> 128 pixels per workgroup, 2 -> 256 (each A fragment multiplies two pixel blocks; the
// launcher takes it where 128-pixel tiles would leave a half-empty last round of workgroups)
template <int PRO, int NT>      // PRO: 0 = no prologue, 1 = GroupNorm apply, 2 = + SiLU
__global__ void __launch_bounds__(512, 2) conv_pw_kernel(const PwParams p) {
    constexpr int PX = PW_PX * NT, XBUF = PW_CIC * PX;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const X = smem;                          // [2][32 channels][PX pixels]
    float* const gnL = smem + 2 * XBUF;             // [2][Cin]: scale, shift of this image (PRO != 0)

    int work;
    {   // XCD-aware bijective remap (conv_mfma.hip)
        const int L = blockIdx.x, nwg = p.nwg;
        const int xcd = L & 7, slot = L >> 3, q = nwg >> 3, r = nwg & 7;
        work = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int co_t = work % p.n_co_tiles;
    const int t = work / p.n_co_tiles;
    const int pt = t % p.n_px_tiles, b = t / p.n_px_tiles;
    const int px0 = pt * PX, co0 = co_t * PW_CO;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave_u >> 2, wn = wave_u & 3;
    const int Cin = p.c0 + p.c1, HW = p.HW, n = p.nchunks;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in0 + (size_t)b * p.c0 * HW), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in1 ? p.in1 + (size_t)b * p.c1 * HW : p.in0), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpw), 0, -1, 0x00020000);

    // staging plan: thread = (4 consecutive pixels, channel ch_lo + CSTEP i of the chunk), i = 0 .. XE-1
    constexpr int TPR = PX / 4;                               // threads per channel row (32 | 64)
    constexpr int CSTEP = 512 / TPR, XE = PW_CIC / CSTEP;     // channels per pass (16 | 8), passes (2 | 4)
    const int px4 = tid % TPR, ch_lo = tid / TPR;
    unsigned voff[XE];
#pragma unroll
    for (int i = 0; i < XE; ++i) voff[i] = 4u * (unsigned)((ch_lo + CSTEP * i) * HW + px0 + 4 * px4);
    const int x_st = ch_lo * PX + 4 * px4;                    // LDS float index of element 0; element i is CSTEP i channels on
    const unsigned u_lane = 16u * (unsigned)lane;
    const int b_base = half * PX + 32 * NT * wn + l31;

    pw_f32x16 acc[NT];
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t2][r] = 0.0f;
    pw_v4f xr[XE], wr[4];

    // (every load is unconditional at a clamped chunk index: the compiler then counts the outstanding loads exactly)
    auto load_x = [&](int c) {
        const int cc0 = PW_CIC * min(c, n - 1);
        const bool first = cc0 < p.c0;                         // the concat seam lies on a chunk boundary (launcher)
        const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
        const unsigned soff = 4u * (unsigned)((first ? cc0 : cc0 - p.c0) * HW);
#pragma unroll
        for (int i = 0; i < XE; ++i) xr[i] = __builtin_bit_cast(pw_v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], soff, 0));
    };
    auto load_w = [&](int c) {
        const unsigned s0 = 1024u * (unsigned)(4 * min(c, n - 1) * p.n_co32 + co_t * 2 + wm);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            wr[q] = __builtin_bit_cast(pw_v4f, __builtin_amdgcn_raw_buffer_load_b128(rsw, u_lane, s0 + 1024u * (unsigned)(q * p.n_co32), 0));
    };
    auto stage_x = [&](const int buf, int c) {
        float* dst = X + buf * XBUF + x_st;
#pragma unroll
        for (int i = 0; i < XE; ++i) {
            pw_v4f v = xr[i];
            if constexpr (PRO != 0) {
                const int ch = PW_CIC * min(c, n - 1) + ch_lo + CSTEP * i;
                const float sc = gnL[ch], sh = gnL[Cin + ch];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float e = v[k] * sc + sh;
                    if constexpr (PRO == 2) e = pw_silu(e);
                    v[k] = e;
                }
            }
            *reinterpret_cast<pw_v4f*>(dst + i * CSTEP * PX) = v;
        }
    };
    auto mfma_chunk = [&](const int buf) {
        const float* Bm = X + buf * XBUF + b_base;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2)
                    acc[t2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[q][s], Bm[(8 * q + 2 * s) * PX + 32 * t2], acc[t2], 0, 0, 0);
    };

    load_x(0);
    load_w(0);
    if constexpr (PRO != 0) {
        for (int i = tid; i < Cin; i += 512) {
            gnL[i] = p.gn_scale[(size_t)b * Cin + i];
            gnL[Cin + i] = p.gn_shift[(size_t)b * Cin + i];
        }
        __syncthreads();
    }
    stage_x(0, 0);
    load_x(1);
    __syncthreads();
    auto step = [&](const int c, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        mfma_chunk(BUF);
        load_w(c + 1);
        stage_x(BUF ^ 1, c + 1);
        load_x(c + 2);
        __syncthreads();
    };
    {
        int c = 0;
        for (; c + 1 < n; c += 2) {
            step(c, std::integral_constant<int, 0>{});
            step(c + 1, std::integral_constant<int, 1>{});
        }
        if (c < n) step(c, std::integral_constant<int, 0>{});
    }

    // ---- epilogue: bias + per-sample channel bias + residual, NCHW store; GroupNorm partials of what was stored
    const int slots = p.n_px_tiles * 4 * NT;
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2) {
    const size_t pix = (size_t)px0 + 32 * (NT * wn + t2) + l31;
    const int slot = (pt * 4 + wn) * NT + t2;
    // (all residual / bias operands of the block are requested before the first is used)
    float add[16], res[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int coc = min(co0 + 32 * wm + 8 * (r >> 2) + 4 * half + (r & 3), p.Cout - 1);
        add[r] = 0.0f;
        if (p.bias) add[r] += p.bias[coc];
        if (p.chan_bias) add[r] += p.chan_bias[(size_t)b * p.chan_bias_stride + coc];
        res[r] = p.residual ? p.residual[((size_t)b * p.Cout + coc) * HW + pix] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * wm + 8 * (r >> 2) + 4 * half + (r & 3);
        float v = acc[t2][r] + add[r] + res[r];
        if (p.relu) v = fmaxf(v, 0.0f);
        if (co < p.Cout) p.out[((size_t)b * p.Cout + co) * HW + pix] = v;
        if (p.stats) {
            const float s1 = pw_half_wave_sum(v);
            const float d = v - s1 * (1.0f / 32.0f);
            const float q = pw_half_wave_sum(d * d);
            if (l31 == 0 && co < p.Cout)
                reinterpret_cast<float4*>(p.stats)[((size_t)b * p.Cout + co) * slots + slot] = make_float4(32.0f, s1, q, 0.0f);
        }
    }
    }
}

// The shapes this kernel takes (everything else stays on the generic kernel): whole tiles, whole chunks, the concat seam on a
// chunk boundary, 16-byte aligned planes, 32-bit byte offsets inside an image of either source and inside the filter tensor.
bool conv_pointwise_applicable(const sisic_conv_args& a) {
    if (a.ksize != 1 || a.stride != 1 || a.upsample) return false;
    const int HW = a.Hin * a.Win, Cin = a.c0 + a.c1;
    if (HW % PW_PX != 0 || Cin % PW_CIC != 0 || (a.c1 != 0 && a.c0 % PW_CIC != 0)) return false;
    if (((reinterpret_cast<uintptr_t>(a.in0) | reinterpret_cast<uintptr_t>(a.in1)) & 15) != 0) return false;
    if (4.0 * std::max(a.c0, a.c1) * HW >= 4294967296.0 || 4.0 * Cin * conv_cout_pad(a.Cout) >= 4294967296.0) return false;
    if (a.gn_scale && Cin > 2048) return false;               // GroupNorm operands of one image staged in LDS
    return true;
}

int conv_pointwise_stats_slots(const sisic_conv_args& a) { return a.Hin * a.Win / 32; }       // one slot per 32 pixels, either tile

template <int PRO, int NT>
static int launch_pw(sisic_ctx* ctx, PwParams& p, int Cin, hipStream_t s) {
    constexpr int PX = PW_PX * NT;
    p.n_px_tiles = p.HW / PX;
    const int64_t nwg = (int64_t)p.B * p.n_px_tiles * p.n_co_tiles;
    SISIC_REQUIRE(nwg > 0 && nwg < (int64_t(1) << 31), "conv2d(pointwise): grid too large");
    p.nwg = (int)nwg;
    const size_t lds = sizeof(float) * (size_t)(2 * PW_CIC * PX + (PRO ? 2 * Cin : 0));
    static std::atomic<uint64_t> opt{0};
    auto kern = conv_pw_kernel<PRO, NT>;
    SISIC_TRY(ensure_dynamic_lds(ctx, reinterpret_cast<const void*>(kern), (int)lds, opt));
    hipLaunchKernelGGL(kern, dim3(p.nwg), dim3(512), lds, s, p);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

int launch_conv_pointwise(sisic_ctx* ctx, const sisic_conv_args& a, hipStream_t s) {
    SISIC_REQUIRE(conv_pointwise_applicable(a), "conv2d(pointwise): shape not supported by tile_cfg 20");
    PwParams p{};
    const int Cin = a.c0 + a.c1;
    p.in0 = a.in0; p.in1 = a.in1; p.c0 = a.c0; p.c1 = a.c1; p.B = a.B; p.HW = a.Hin * a.Win;
    p.wpw = a.w_packed + (size_t)conv_cin_pad(Cin, 1) * conv_cout_pad(a.Cout);     // the second layout (pack_device.h)
    p.n_co32 = conv_cout_pad(a.Cout) / 32;
    p.bias = a.bias; p.Cout = a.Cout;
    p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift;
    p.chan_bias = a.chan_bias; p.chan_bias_stride = a.chan_bias_stride; p.residual = a.residual; p.relu = a.relu;
    p.out = a.out; p.stats = a.stats_out;
    p.n_co_tiles = cdiv(a.Cout, PW_CO);
    p.nchunks = Cin / PW_CIC;
    const int pro = a.gn_scale == nullptr ? 0 : (a.gn_silu ? 2 : 1);
    // (256-pixel tiles, NT = 2, were measured for q/k/v at 16x16 -- 1536 workgroups of 128 pixels on 1024 places -- and are
    //  slower: 100 vs 75 us; 64 KB of LDS halve the resident workgroups.  profiles/r03/conv_bench_pointwise.txt)
    if (pro == 2) return launch_pw<2, 1>(ctx, p, Cin, s);
    if (pro == 1) return launch_pw<1, 1>(ctx, p, Cin, s);
    return launch_pw<0, 1>(ctx, p, Cin, s);
}

}  // namespace sisic
