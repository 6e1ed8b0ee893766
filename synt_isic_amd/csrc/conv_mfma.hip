// conv_mfma.hip -- fp32 NCHW convolution as an implicit GEMM on the gfx950 f32 MFMA pipe.
//
// Replaces every F.conv2d instance on the hot path (52 conv3x3 + 14 conv1x1 shortcuts per
// UNet forward, plus the attention q/k/v/out projections which are 1x1 convolutions in
// NCHW) together with the elementwise work diffusers runs around them -- GroupNorm-apply,
// SiLU, torch.cat of the skip tensor, nearest 2x upsampling, the time-embedding add and
// the residual add.
//
// GEMM view:  D[co, pixel] = sum_k W[co, k] * X[k, pixel],  k = (ci, ky, kx).
//   * pixels ride the MFMA lane dimension, so NCHW rows are read and written as
//     contiguous 128-byte segments and no layout change is needed at the boundary;
//   * v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, one VGPR per operand; the two k of an
//     instruction are two consecutive input channels at the same filter tap;
//   * a workgroup owns CO_TILE=64 output channels x PIX pixels (a TH x TW rectangle of
//     one image) and walks the input channels in chunks of CIC, double-buffered in LDS:
//     the input halo tile [CIC][IH][IW] (GroupNorm+SiLU applied on the way in, zero
//     padding after it) and the weight slab [CIC][k*k][64];
//   * operands are read from LDS with ds_read_b32 at compile-time offsets from one
//     per-lane base each; per k-step a wave issues MT+NT reads for MT*NT MFMAs.
//
// Algorithmic bytes per launch: 4*B*(Cin*Hin*Win + Cout*Hout*Wout)
//   + 4*(Cin*Cout*k*k + Cout) (+ 4*B*Cout*Hout*Wout when a residual is read).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "pack_device.h"

namespace sisic {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* in0;
    const float* in1;
    int c0, c1;
    int B, Hin, Win;   // source tensors
    int Hc, Wc;        // conv input extent (after the optional 2x upsample)
    int Hout, Wout;
    int ups;
    int zero_insert;   // with ups: the 2x grid holds the source at even coordinates and zeros elsewhere (transposed stride-2 conv)
    const float* w;    // packed [Cin_pad][KK][cout_pad]
    int cout_pad;
    const float* bias;
    int Cout;
    const float* gn_scale;
    const float* gn_shift;
    int gn_silu;
    const float* chan_bias;
    int chan_bias_stride;
    const float* residual;
    int relu;
    float* out;
    float* stats;      // optional GroupNorm partials [B][Cout][tiles_y*tiles_x*WN][4] = (count, sum, centred M2, 0)
    int* slots_query;  // host only: when set, launch_cfg reports the slot count of its tiling instead of launching
    int tiles_x, tiles_y, n_co_tiles, nwg, nchunks;
};

// sum over each aligned run of 32 lanes (the pixels of one accumulator row): DPP butterflies inside the two rows of 16,
// then one cross-row exchange.  Fixed order: the GroupNorm partials are bit-reproducible.
__device__ __forceinline__ float half_wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v + __shfl_xor(v, 16);
}

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }   // v_rcp_f32, not the IEEE division sequence

template <int KS, int STRIDE, int MT, int NT, int WM, int WN, int TW, int CIC, int KSP = 1>
struct ConvGeom {
    static constexpr int NWAVES = WM * WN * KSP;              // KSP > 1: wave groups split the channel pairs of a chunk
    static constexpr int NTHR = 64 * NWAVES;
    static constexpr int CO_TILE = WM * MT * 32;
    static constexpr int PIX = WN * NT * 32;
    static constexpr int TH = PIX / TW;
    static constexpr int KK = KS * KS;
    static constexpr int PAD = KS / 2;
    static constexpr int IH = (TH - 1) * STRIDE + KS;
    static constexpr int IW = (TW - 1) * STRIDE + KS;
    static constexpr int TPC = NTHR / CIC;                    // threads staging one channel
    static constexpr int EPT = (IH * IW + TPC - 1) / TPC;     // input elements per thread per chunk
    static constexpr int CHS = EPT * TPC;                     // LDS floats per staged channel (>= IH*IW, padded so
                                                              // every thread stores all its EPT elements)
    static constexpr int IN_BUF = ((CIC * CHS + 3) / 4) * 4;  // floats, 16-B multiple
    static constexpr int W_F4 = CIC * KK * CO_TILE / 4;       // float4 per weight slab
    static constexpr int W_F4_PT = (W_F4 + NTHR - 1) / NTHR;
    static constexpr int W_BUF = W_F4_PT * NTHR * 4;          // padded likewise
    static constexpr size_t LDS_BYTES = size_t(2) * (IN_BUF + W_BUF) * sizeof(float);
    static_assert(CO_TILE == CONV_CO_TILE, "weight packing assumes 64-channel tiles");
    static_assert(PIX % TW == 0, "tile must be whole rows");
    static_assert(NTHR % CIC == 0, "staging split");
    static_assert(CIC % (2 * KSP) == 0, "two channels per MFMA, whole pairs per K group");
};

// V4 (1x1 stride 1, H*W a multiple of 4): the pixel tile of a channel is one contiguous run, so a staging thread moves
// four consecutive pixels per instruction (global_load_dwordx4 -> ds_write_b128) instead of single elements strided by
// the staging-thread count: 5x fewer vector instructions per chunk and 256-byte instead of 64-byte global segments.
template <int KS, int STRIDE, int MT, int NT, int WM, int WN, int TW, int CIC, int OCC, int KSP, bool V4 = false>
__global__ void __launch_bounds__(64 * WM * WN * KSP, OCC) conv_mfma_kernel(const ConvParams p) {
    using G = ConvGeom<KS, STRIDE, MT, NT, WM, WN, TW, CIC, KSP>;
    static_assert(!V4 || (KS == 1 && STRIDE == 1 && TW == G::PIX && G::EPT % 4 == 0 && G::CHS % 4 == 0),
                  "the float4 staging path is for flat 1x1 tiles");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const in_lds = smem;                    // [2][IN_BUF]
    float* const w_lds = smem + 2 * G::IN_BUF;     // [2][W_BUF]

    // ---- which tile: XCD-aware bijective remap (blocks b, b+8 share an XCD's L2) so that the
    // co-tiles of one pixel tile and the neighbouring pixel tiles of one image land on one XCD.
    int work;
    {
        const int L = blockIdx.x, nwg = p.nwg;
        const int xcd = L & 7, slot = L >> 3, q = nwg >> 3, r = nwg & 7;
        work = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int co_t = work % p.n_co_tiles;
    int tile = work / p.n_co_tiles;
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int ty = tile % p.tiles_y;
    const int b = tile / p.tiles_y;
    const int oy0 = ty * G::TH, ox0 = tx * TW;   // output-space origin of the tile
    const int co0 = co_t * G::CO_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kg = wave / (WM * WN), wq = wave % (WM * WN);   // K-split group, position inside the output tile
    const int wm = wq / WN, wn = wq % WN;
    const int half = lane >> 5, l31 = lane & 31;

    // ---- input staging plan (invariant over the channel loop)
    const int sci = tid / G::TPC;   // staged channel within the chunk
    const int sl = tid % G::TPC;
    const int HWin = p.Hin * p.Win;
    const int Cin = p.c0 + p.c1;
    int goff[G::EPT];
    unsigned vmask = 0;
    constexpr int EPT4 = V4 ? G::EPT / 4 : 1;
    int goff4[EPT4];              // V4: first pixel of this thread's i-th float4 (clamped), valid bit i in vmask4
    unsigned vmask4 = 0;
    if constexpr (V4) {
#pragma unroll
        for (int i = 0; i < EPT4; ++i) {
            const int px = ox0 + 4 * sl + i * G::TPC * 4;
            const bool v = px < p.Wc;                         // H*W % 4 == 0: a float4 is inside or outside as a whole
            goff4[i] = v ? px : 0;
            vmask4 |= (v ? 1u : 0u) << i;
        }
    }
#pragma unroll
    for (int i = 0; i < (V4 ? 0 : G::EPT); ++i) {
        const int e = sl + i * G::TPC;
        const int yy = e / G::IW, xx = e % G::IW;
        const int gy = oy0 * STRIDE - G::PAD + yy;
        const int gx = ox0 * STRIDE - G::PAD + xx;
        const bool v = (e < G::IH * G::IW) && gy >= 0 && gy < p.Hc && gx >= 0 && gx < p.Wc &&
                       !(p.zero_insert && ((gy | gx) & 1));
        goff[i] = v ? ((gy >> p.ups) * p.Win + (gx >> p.ups)) : 0;
        vmask |= (v ? 1u : 0u) << i;
    }
    static_assert(G::EPT <= 32, "valid mask is 32 bits");

    // ---- MFMA operand bases
    const int a_base = half * G::KK * G::CO_TILE + wm * MT * 32 + l31;
    int b_base;
    {
        const int pix = wn * NT * 32 + l31;
        const int y = pix / TW, x = pix % TW;
        b_base = half * G::CHS + (y * STRIDE) * G::IW + x * STRIDE;
    }
    // offset of N-tile nt relative to nt=0 (32 pixels further along the tile)
    constexpr int NT_STEP = (TW >= 32) ? ((TW > 32) ? 32 * STRIDE : STRIDE * G::IW) : (32 / TW) * STRIDE * G::IW;
    static_assert(TW >= 32 || 32 % TW == 0, "TW must divide 32");
    static_assert(TW <= 32 || TW % 32 == 0, "TW must be a multiple of 32");

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    float rin[G::EPT];
    float rw[G::W_F4_PT * 4];   // scalars, not a float4 array: hipcc keeps a float4[] staging array in scratch
    float gsc = 1.0f, gsh = 0.0f;
    bool cval = false;

    // All staging loads are UNCONDITIONAL at clamped, always-valid addresses and masked afterwards:
    // a per-element "load or zero" select makes hipcc branch around every load and wait vmcnt(0)
    // inside each branch, which serialises the whole prefetch (cdna_hip_programming.md, .s-level trap c).
    const int prologue = (p.gn_scale == nullptr) ? 0 : (p.gn_silu ? 2 : 1);   // wave-uniform

    // (always_inline: called from two instances of `step`, the 7x7 instance's compute_chunk was left as a function of its own --
    //  its accumulators went through memory and config 5 ran four times slower until tools/bench_configs.py was run again)
    auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
        const int c = chunk * CIC + sci;
        cval = c < Cin;
        const int cc = min(c, Cin - 1);
        const float* src = (cc < p.c0) ? p.in0 + ((size_t)b * p.c0 + cc) * HWin
                                       : p.in1 + ((size_t)b * p.c1 + (cc - p.c0)) * HWin;
        if constexpr (V4) {
#pragma unroll
            for (int i = 0; i < EPT4; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(src + goff4[i]);
                rin[4 * i + 0] = t.x; rin[4 * i + 1] = t.y; rin[4 * i + 2] = t.z; rin[4 * i + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < G::EPT; ++i) rin[i] = src[goff[i]];
        }
        if (prologue) {
            gsc = p.gn_scale[(size_t)b * Cin + cc];
            gsh = p.gn_shift[(size_t)b * Cin + cc];
        }
        const float* wsrc = p.w + (size_t)chunk * CIC * G::KK * p.cout_pad + co0;
#pragma unroll
        for (int i = 0; i < G::W_F4_PT; ++i) {
            const int f = min(tid + i * G::NTHR, G::W_F4 - 1);
            const int rr = f / (G::CO_TILE / 4), c4 = f % (G::CO_TILE / 4);
            const float4 t = *reinterpret_cast<const float4*>(wsrc + (size_t)rr * p.cout_pad + c4 * 4);
            rw[4 * i + 0] = t.x; rw[4 * i + 1] = t.y; rw[4 * i + 2] = t.z; rw[4 * i + 3] = t.w;
        }
    };

    // (buf is a compile-time constant in the channel loop below: no buffer-parity arithmetic in the operand addresses)
    auto store_chunk = [&](const int buf) __attribute__((always_inline)) {
        float* dst = in_lds + buf * G::IN_BUF + sci * G::CHS + sl;
        const unsigned m = cval ? vmask : 0u;       // padding / missing channels stay zero AFTER the prologue
        float v[G::EPT];
        if (prologue == 2) {
#pragma unroll
            for (int i = 0; i < G::EPT; ++i) v[i] = silu_f(rin[i] * gsc + gsh);
        } else if (prologue == 1) {
#pragma unroll
            for (int i = 0; i < G::EPT; ++i) v[i] = rin[i] * gsc + gsh;
        } else {
#pragma unroll
            for (int i = 0; i < G::EPT; ++i) v[i] = rin[i];
        }
        if constexpr (V4) {
            float* dst4 = in_lds + buf * G::IN_BUF + sci * G::CHS + 4 * sl;
            const unsigned m4 = cval ? vmask4 : 0u;
#pragma unroll
            for (int i = 0; i < EPT4; ++i) {
                const bool on = (m4 >> i) & 1u;
                *reinterpret_cast<float4*>(dst4 + i * G::TPC * 4) =
                    make_float4(on ? v[4 * i + 0] : 0.0f, on ? v[4 * i + 1] : 0.0f, on ? v[4 * i + 2] : 0.0f, on ? v[4 * i + 3] : 0.0f);
            }
        } else {
#pragma unroll
            for (int i = 0; i < G::EPT; ++i) dst[i * G::TPC] = ((m >> i) & 1u) ? v[i] : 0.0f;
        }
        float* wdst = w_lds + buf * G::W_BUF + tid * 4;
#pragma unroll
        for (int i = 0; i < G::W_F4_PT; ++i)
            *reinterpret_cast<float4*>(wdst + i * G::NTHR * 4) =
                make_float4(rw[4 * i + 0], rw[4 * i + 1], rw[4 * i + 2], rw[4 * i + 3]);
    };

    auto compute_chunk = [&](const int buf) __attribute__((always_inline)) {
        const float* A = w_lds + buf * G::W_BUF + a_base;
        const float* Bm = in_lds + buf * G::IN_BUF + b_base;
#pragma unroll
        for (int cp = kg * (CIC / 2 / KSP); cp < (kg + 1) * (CIC / 2 / KSP); ++cp) {
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    float a[MT], bb[NT];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        a[m] = A[(2 * cp * G::KK + ky * KS + kx) * G::CO_TILE + m * 32];
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        bb[n] = Bm[2 * cp * G::CHS + ky * G::IW + kx + n * NT_STEP];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[n], acc[m][n], 0, 0, 0);
                }
            }
        }
    };

    // ---- channel loop: register-staged double buffering, one barrier per chunk
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    auto step = [&](const int chunk, auto buf_tag) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value;
        const bool more = chunk + 1 < p.nchunks;
        if (more) load_chunk(chunk + 1);
        compute_chunk(BUF);
        if (more) store_chunk(BUF ^ 1);
        __syncthreads();
    };
    {
        int chunk = 0;
        for (; chunk + 1 < p.nchunks; chunk += 2) {
            step(chunk, std::integral_constant<int, 0>{});
            step(chunk + 1, std::integral_constant<int, 1>{});
        }
        if (chunk < p.nchunks) step(chunk, std::integral_constant<int, 0>{});
    }

    // ---- K-split: the wave groups hold partial sums over disjoint channel pairs of the same output tile; groups
    // 1.. hand their accumulators to group 0 through LDS (the staging buffers are dead after the loop's last barrier).
    if constexpr (KSP > 1) {
        static_assert(KSP == 2, "two K groups");
        static_assert((size_t)WM * WN * MT * NT * 16 * 64 * sizeof(float) <= G::LDS_BYTES, "reduction slab fits the staging LDS");
        float* red = smem + (size_t)wq * MT * NT * 16 * 64 + lane;
        if (kg == 1) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((m * NT + n) * 16 + r) * 64] = acc[m][n][r];
        }
        __syncthreads();
        if (kg != 0) return;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] += red[((m * NT + n) * 16 + r) * 64];
    }

    // ---- epilogue: bias + per-sample channel bias (time embedding) + residual, NCHW store.
    // Loads use clamped indices and are unconditional; only the stores are predicated.
    const size_t HWout = (size_t)p.Hout * p.Wout;
    float cbias[MT][16];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            cbias[m][r] = 0.0f;
        }
    }
    if (p.bias) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                cbias[m][r] += p.bias[min(co0 + wm * MT * 32 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, p.Cout - 1)];
    }
    if (p.chan_bias) {
        const float* cb = p.chan_bias + (size_t)b * p.chan_bias_stride;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                cbias[m][r] += cb[min(co0 + wm * MT * 32 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, p.Cout - 1)];
    }
    float psum[MT][16];      // per output channel: sum of this lane's stored values (GroupNorm partials)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) psum[m][r] = 0.0f;
    unsigned pvmask = 0;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int pix = wn * NT * 32 + n * 32 + l31;
        const int oy = oy0 + pix / TW, ox = ox0 + pix % TW;
        const bool pv = oy < p.Hout && ox < p.Wout;
        pvmask |= pv ? (1u << n) : 0u;
        const size_t pix_off = (size_t)min(oy, p.Hout - 1) * p.Wout + min(ox, p.Wout - 1);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float res[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) res[r] = 0.0f;
            if (p.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = min(co0 + wm * MT * 32 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, p.Cout - 1);
                    res[r] = p.residual[((size_t)b * p.Cout + co) * HWout + pix_off];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * MT * 32 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[m][n][r] + cbias[m][r] + res[r];
                if (p.relu) v = fmaxf(v, 0.0f);
                if (pv && co < p.Cout) p.out[((size_t)b * p.Cout + co) * HWout + pix_off] = v;
                acc[m][n][r] = v;
                psum[m][r] += pv ? v : 0.0f;
            }
        }
    }

    // ---- GroupNorm partials of what was just stored: per output channel and wave, (count, sum, sum of squared
    // deviations from the wave's own mean) over the wave's NT*32 pixels; merged exactly by gn_finalize_kernel.
    if (p.stats) {
        const float cnt = half_wave_sum((float)__builtin_popcount(pvmask));
        const float inv = 1.0f / fmaxf(cnt, 1.0f);
        const int slots = p.tiles_x * p.tiles_y * WN, slot = (ty * p.tiles_x + tx) * WN + wn;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * MT * 32 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float s1 = half_wave_sum(psum[m][r]);
                const float mean = s1 * inv;
                float q = 0.0f;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float d = acc[m][n][r] - mean;
                    q += ((pvmask >> n) & 1u) ? d * d : 0.0f;
                }
                q = half_wave_sum(q);
                if (l31 == 0 && co < p.Cout)
                    reinterpret_cast<float4*>(p.stats)[((size_t)b * p.Cout + co) * slots + slot] = make_float4(cnt, s1, q, 0.0f);
            }
        }
    }
}


// OIHW -> [Cin_pad][KK][cout_pad], zero padded
__global__ void conv_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int KK, int cin_pad, int cout_pad,
                                 float* __restrict__ out) {
    const size_t first = (size_t)cin_pad * KK * cout_pad;
    const size_t total = KK == 1 ? 3 * first + first / 2 : first;                   // 1x1: all three layouts (pack_device.h)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        conv_pack_elem(i, w, Cout, Cin, KK, cin_pad, cout_pad, out);
    }
}

template <int KS, int STRIDE, int MT, int NT, int WM, int WN, int TW, int CIC, int OCC = 2, int KSP = 1>
static int launch_cfg(sisic_ctx* ctx, ConvParams& p, hipStream_t s) {
    using G = ConvGeom<KS, STRIDE, MT, NT, WM, WN, TW, CIC, KSP>;
    p.tiles_x = cdiv(p.Wout, TW);
    p.tiles_y = cdiv(p.Hout, G::TH);
    p.n_co_tiles = p.cout_pad / G::CO_TILE;
    p.nchunks = cdiv(p.c0 + p.c1, CIC);
    const int64_t nwg = (int64_t)p.B * p.tiles_x * p.tiles_y * p.n_co_tiles;
    SISIC_REQUIRE(nwg > 0 && nwg < (int64_t(1) << 31), "conv2d: grid of %lld workgroups unsupported", (long long)nwg);
    p.nwg = (int)nwg;
    if (p.slots_query) {           // sisic_conv_stats_slots(): report the partial-statistics layout, launch nothing
        *p.slots_query = p.tiles_x * p.tiles_y * WN;
        return SISIC_OK;
    }
    if constexpr (KS == 1 && STRIDE == 1 && TW == G::PIX && G::EPT % 4 == 0) {
        // flat 1x1 tile: float4 staging when every plane is a whole number of 16-byte aligned float4
        const bool aligned = ((reinterpret_cast<uintptr_t>(p.in0) | reinterpret_cast<uintptr_t>(p.in1)) & 15) == 0;
        if (p.Wc % 4 == 0 && aligned) {
            auto kern4 = conv_mfma_kernel<KS, STRIDE, MT, NT, WM, WN, TW, CIC, OCC, KSP, true>;
            static std::atomic<uint64_t> lds_opt_in4{0};     // one bit per device (common.h)
            SISIC_TRY(ensure_dynamic_lds(ctx, reinterpret_cast<const void*>(kern4), (int)G::LDS_BYTES, lds_opt_in4));
            hipLaunchKernelGGL(kern4, dim3(p.nwg), dim3(G::NTHR), G::LDS_BYTES, s, p);
            SISIC_HIP(hipGetLastError());
            return SISIC_OK;
        }
    }
    auto kern = conv_mfma_kernel<KS, STRIDE, MT, NT, WM, WN, TW, CIC, OCC, KSP>;
    static std::atomic<uint64_t> lds_opt_in{0};     // one bit per device (common.h)
    SISIC_TRY(ensure_dynamic_lds(ctx, reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_opt_in));
    hipLaunchKernelGGL(kern, dim3(p.nwg), dim3(G::NTHR), G::LDS_BYTES, s, p);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

int launch_conv_pack(sisic_ctx*, const float* w, int Cout, int Cin, int k, float* packed, hipStream_t s) {
    SISIC_REQUIRE(k == 1 || k == 3 || k == 7, "conv_pack: ksize %d unsupported", k);
    const int cin_pad = conv_cin_pad(Cin, k), cout_pad = conv_cout_pad(Cout);
    const size_t total = (size_t)cin_pad * k * k * cout_pad;
    const int blocks = (int)std::min<size_t>((total * (k == 1 ? 4 : 1) + 255) / 256, 4096);
    hipLaunchKernelGGL(conv_pack_kernel, dim3(blocks), dim3(256), 0, s, w, Cout, Cin, k * k, cin_pad, cout_pad, packed);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// Tile configurations.  id -> (KS, STRIDE, MT, NT, WM, WN, TW):
//   3x3 s1:  1: 2,2,1,4,TW64   2: 2,2,1,4,TW32   3: 2,2,1,4,TW16   4: 1,1,2,2,TW8   5: 2,1,1,4,TW16 (PIX128)
//            6: 2,1,1,4,TW64   7: 2,1,1,4,TW32 (PIX128)   8: 1,2,2,4,TW32   9: 1,2,2,4,TW16 (8 waves, PIX256)
//           14: 1,1,2,4,TW16  15: 1,1,2,4,TW8 (8 waves, PIX128)   50: vector-ALU kernel for Cout <= 4
//           16 / 17: cfg 4's 64x64 tile with two K-split wave groups (8 waves), 16- / 8-channel chunks
//   3x3 s2: 11: 2,1,1,4,TW32  12: 2,1,1,4,TW16  13: 1,1,2,2,TW8   18 / 19: cfg 13's tile with two K-split wave groups, 16- / 8-channel chunks
//   1x1   : 21: 2,2,1,4,TW256 22: 1,1,2,2,TW64  23: 2,1,1,4,TW128
//   1x1 s2: 31: 2,1,1,4,TW32  32: 2,1,1,4,TW16  33: 1,1,2,2,TW8      7x7 s2: 41: 2,1,1,4,TW32 (CIC 4)
static int dispatch_conv2d(sisic_ctx* ctx, const sisic_conv_args& a, hipStream_t s, int* slots_query);

// Winograd F(2x2,3x3) is taken for 3x3 stride-1 convolutions with transformed filters at hand: when forced by
// tile_cfg 60..74 / 78 / 79 / 90 / 91, or automatically from 12x12 outputs up and (K-split form) at 8x8 (per-thread load offsets
// there are 32-bit).  Returns the tile configuration, 0 = not Winograd.
static int winograd_cfg(const sisic_conv_args& a) {
    if (!(a.ksize == 3 && a.stride == 1 && a.w_winograd != nullptr && a.Cout > 4) || a.upsample == 2) return 0;
    if ((a.tile_cfg >= 60 && a.tile_cfg <= 74) || a.tile_cfg == 78 || a.tile_cfg == 79 || (a.tile_cfg >= 90 && a.tile_cfg <= 92)) return a.tile_cfg;
    if (a.tile_cfg != 0) return 0;
    const int Hout = a.Hin << (a.upsample ? 1 : 0), Wout = a.Win << (a.upsample ? 1 : 0);
    const bool fits32 = 16.0 * std::max(a.c0, a.c1) * a.Hin * a.Win < 4294967296.0;
    if (!fits32) return 0;
    if (Hout >= 12 && Wout >= 12) {
        // second geometry (conv_winograd_wide.inc) unless SISIC_WINO_WIDE=0; nearest-2x inputs keep the nine-position form
        static const bool wide_on = [] { const char* e = std::getenv("SISIC_WINO_WIDE"); return !e || std::atoi(e) != 0; }();
        // measured per layer (tools/conv_bench.py, profiles/r02/conv_bench_geometries.txt): the 128-channel form wins 8-12 % on
        // every Cout >= 128 layer, the 64-channel two-workgroups-per-CU form 1-10 % on every Cout <= 64 layer
        // fp32-equivalent products on the bf16 matrix pipe (conv_winograd_bf3.inc) unless SISIC_WINO_BF16X3=0: 64 channels x
        // 16 x 16 pixels per workgroup, so only where those tiles are (nearly) full; measured 1.36 - 1.43x the third f32 form on every
        // such layer of the headline model (profiles/r03/conv_bench_bf16x3.txt), the same error against float64
        static const bool bf3_on = [] { const char* e = std::getenv("SISIC_WINO_BF16X3"); return !e || std::atoi(e) != 0; }();
        // (ragged planes too when at least three quarters of the 16x16-pixel tiles' area is inside: the classifier's 56 / 28 / 14)
        const int th = (Hout + 15) / 16 * 16, tw = (Wout + 15) / 16 * 16;
        // (nearest-2x inputs as well: all 16 positions on the bf16 pipe beat the f32 form's 9 on every upsample layer of the
        //  UNet -- 562 -> 512 us over the three, profiles/r03/conv_bench_bf16x3.txt)
        if (bf3_on && a.Cout % 64 == 0 && 4 * Hout * Wout >= 3 * th * tw && a.c0 + a.c1 >= 16 &&
            a.c0 + a.c1 <= 2048 &&                                      // (its LDS table of the image's GroupNorm operands)
            4.0 * a.Cout * Hout * Wout < 2147483648.0)
            return 74;
        if (wide_on && !a.upsample) return a.Cout > 64 ? 68 : 69;
        return 66;
    }
    // the 8x8 level (and the classifier's 7x7): four images per workgroup and the input channels split four ways
    // keep all CUs busy
    const int Cin = a.c0 + a.c1;
    static const bool ksplit_on = [] { const char* e = std::getenv("SISIC_KSPLIT"); return !e || std::atoi(e) != 0; }();
    // (no batch-size condition anywhere in this function: an image's bits must not depend on the batch it is in)
    if (ksplit_on && Hout <= 8 && Wout <= 8 && Hout >= 5 && Wout >= 5 && Cin >= 128 && Cin % 32 == 0 && a.Cout >= 128) {
        // second geometry with two images per workgroup (tile_cfg 91): 45 -> 39 us and 67 -> 57 us per launch at B = 64
        static const bool wide_on = [] { const char* e = std::getenv("SISIC_WINO_WIDE"); return !e || std::atoi(e) != 0; }();
        // tile_cfg 92: the same split with fp32-equivalent products on the bf16 pipe, four images per workgroup
        // (conv_winograd_bf3.inc): 38 -> 32 and 59 -> 47 us per launch at B = 64 (profiles/r03/conv_bench_bf16x3.txt)
        static const bool bf3_on = [] { const char* e = std::getenv("SISIC_WINO_BF16X3"); return !e || std::atoi(e) != 0; }();
        if (bf3_on && !a.upsample && a.Cout % 64 == 0 && Cin <= 512) return 92;
        return (wide_on && !a.upsample) ? 91 : 90;
    }
    return 0;
}
static bool winograd_selected(const sisic_conv_args& a) { return winograd_cfg(a) != 0; }

// GroupNorm partials (sisic_conv_args.stats_out): the Winograd kernels' output transform leaves one slot per
// workgroup tile of an image (16x16 outputs, or 8x8 for the four-image tilings; the K-split form one per image from
// its reduction); the direct MFMA kernel one per pixel tile and pixel-wave (the dispatch below is asked which tiling
// it would launch).  The vector-ALU small-Cout kernel does not produce them.
int conv_stats_slots(const sisic_conv_args& a) {
    if (const int cfg = winograd_cfg(a)) {
        if (cfg >= 90 && cfg <= 92) return 1;
        const int Hout = a.Hin << (a.upsample ? 1 : 0), Wout = a.Win << (a.upsample ? 1 : 0);
        if ((cfg == 78 || cfg == 79) && wino_latency_ksplit(a.Cout, a.c0 + a.c1, Hout, Wout) > 1) return wino_latency_segments(Hout, Wout);   // from the plane reduction
        if ((cfg >= 68 && cfg <= 73) || cfg == 78 || cfg == 79) return ((Hout + 7) / 8) * ((Wout + 15) / 16);   // 8 x 16 output pixels per workgroup
        const int edge = (cfg == 61 || cfg == 63 || cfg == 65 || cfg == 67) ? 8 : 16;
        return ((Hout + edge - 1) / edge) * ((Wout + edge - 1) / edge);
    }
    int slots = 0;
    sisic_conv_args q = a;
    q.stats_out = nullptr;
    if (dispatch_conv2d(nullptr, q, nullptr, &slots) != SISIC_OK) return 0;
    return slots;
}

// sisic_conv_finalizes(): where a workgroup holds whole GroupNorm groups (eight channels) of an image it finalizes them: the
// K-split 8x8-level forms in their reduction kernel (16 whole channel planes per workgroup: conv_winograd.hip, ReduceFin), the
// bf16x3 Winograd kernel where one tile is the whole image (16x16 and smaller: conv_winograd_bf3.inc)
bool conv_finalizes(const sisic_conv_args& a) {
    if (!a.fin_gamma || a.fin_groups <= 0) return false;
    const int cfg = winograd_cfg(a);
    const int Hout = a.Hin << (a.upsample ? 1 : 0), Wout = a.Win << (a.upsample ? 1 : 0);
    if (a.Cout % a.fin_groups != 0 || a.Cout / a.fin_groups != 8) return false;
    // the bf16x3 Winograd kernel where ONE 16x16-pixel tile is the whole image: a workgroup holds 64 channels of an image whole
    if (cfg == 74) return !a.upsample && Hout <= 16 && Wout <= 16 && a.Cout % 64 == 0;
    if (cfg < 90 || cfg > 92) return false;
    if (Hout * Wout != 64 || a.Cout % 16 != 0) return false;
    // (the 16-byte form of the reduction; its scratch slabs are the library's own allocation)
    return ((reinterpret_cast<uintptr_t>(a.residual) | reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(a.stats_out)) & 15) == 0;
}

int launch_conv2d(sisic_ctx* ctx, const sisic_conv_args& a, hipStream_t s) { return dispatch_conv2d(ctx, a, s, nullptr); }

static int dispatch_conv2d(sisic_ctx* ctx, const sisic_conv_args& a, hipStream_t s, int* slots_query) {
    SISIC_REQUIRE(a.in0 && a.w_packed && a.out, "conv2d: null tensor");
    SISIC_REQUIRE(a.B > 0 && a.Hin > 0 && a.Win > 0 && a.c0 > 0 && a.c1 >= 0 && a.Cout > 0, "conv2d: bad shape");
    SISIC_REQUIRE((a.c1 == 0) == (a.in1 == nullptr), "conv2d: in1/c1 mismatch");
    SISIC_REQUIRE(a.ksize == 1 || a.ksize == 3 || a.ksize == 7, "conv2d: ksize %d unsupported", a.ksize);
    SISIC_REQUIRE(a.stride == 1 || a.stride == 2, "conv2d: stride %d unsupported", a.stride);
    SISIC_REQUIRE(a.ksize != 7 || a.stride == 2, "conv2d: 7x7 is built for stride 2 only (the ResNet stem)");
    SISIC_REQUIRE(!(a.upsample && a.stride != 1), "conv2d: upsample with stride");
    SISIC_REQUIRE((a.gn_scale == nullptr) == (a.gn_shift == nullptr), "conv2d: gn_scale/gn_shift mismatch");

    ConvParams p{};
    p.in0 = a.in0; p.in1 = a.in1; p.c0 = a.c0; p.c1 = a.c1;
    p.B = a.B; p.Hin = a.Hin; p.Win = a.Win;
    p.ups = a.upsample ? 1 : 0;
    p.zero_insert = a.upsample == 2 ? 1 : 0;
    p.Hc = a.Hin << p.ups; p.Wc = a.Win << p.ups;
    const int pad = a.ksize / 2;
    p.Hout = (p.Hc + 2 * pad - a.ksize) / a.stride + 1;
    p.Wout = (p.Wc + 2 * pad - a.ksize) / a.stride + 1;
    p.w = a.w_packed; p.cout_pad = conv_cout_pad(a.Cout);
    p.bias = a.bias; p.Cout = a.Cout;
    p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.gn_silu = a.gn_silu;
    p.chan_bias = a.chan_bias; p.chan_bias_stride = a.chan_bias_stride; p.residual = a.residual; p.relu = a.relu; p.out = a.out;
    p.stats = a.stats_out; p.slots_query = slots_query;

    const int Cin = a.c0 + a.c1;
    const double kk = double(a.ksize) * a.ksize;
    const double out_elems = double(a.B) * a.Cout * p.Hout * p.Wout;
    const double bytes = 4.0 * (double(a.B) * Cin * a.Hin * a.Win + out_elems) + 4.0 * (Cin * a.Cout * kk + a.Cout) +
                         (a.residual ? 4.0 * out_elems : 0.0);
    const double flops = 2.0 * out_elems * Cin * kk;
    int cfg = a.tile_cfg;
    const bool use_wino = winograd_selected(a);
    // F(2x2,3x3): 16 multiplies per 2x2 outputs and channel pair instead of 36
    SISIC_REQUIRE(a.stats_out == nullptr || conv_stats_slots(a) > 0,
                  "conv2d: stats_out given but sisic_conv_stats_slots() is 0 for these arguments");
    // matrix FLOPs actually issued: F(2x2,3x3) multiplies 16 positions per 2x2 outputs instead of 36 taps, and only 9
    // of them for nearest-2x inputs (conv_winograd.hip, upsample form)
    const bool wino_ups9 = use_wino && a.upsample && !a.gn_scale && winograd_cfg(a) == 66;
    ProfileScope prof(slots_query ? nullptr : ctx, s, a.ksize == 1 ? PK_CONV1 : PK_CONV3, bytes, flops,
                      use_wino ? flops * (wino_ups9 ? 9.0 : 16.0) / 36.0 : flops,
                      (use_wino && winograd_cfg(a) == 74) ? PK_WINO_BF3 :
                      (use_wino && !wino_ups9 && (winograd_cfg(a) == 66 || (winograd_cfg(a) >= 68 && winograd_cfg(a) <= 73) || winograd_cfg(a) == 78 || winograd_cfg(a) == 79)) ? PK_WINO_MAIN : -1);

    if (a.ksize == 3 && a.stride == 1 && !a.upsample && a.Cout <= 4 && (cfg == 0 || (cfg >= 50 && cfg <= 52))) {
        if (slots_query) return SISIC_OK;             // no partials from this kernel (slots stay 0)
        return launch_conv_smallcout(ctx, a, s);      // conv_out: vector-ALU kernel, conv_small.hip
    }
    // Winograd F(2x2,3x3): 2.25x fewer MFMA FLOPs.  Auto: from 12x12 output up (64 tiles of one image fill a
    // workgroup); the 8x8 level stays on the direct kernel (too few workgroups of 4 images x 64 channels).
    if (use_wino) {
        SISIC_REQUIRE(!slots_query, "conv2d: internal: slot query on the Winograd path");
        return launch_conv_winograd(ctx, a, a.w_winograd, winograd_cfg(a), s);
    }
    SISIC_REQUIRE((cfg < 60 || cfg > 74) && cfg != 78 && cfg != 79 && (cfg < 90 || cfg > 92), "conv2d: tile_cfg %d needs w_winograd, ksize 3 and stride 1", cfg);
    if (a.ksize == 7) {
        if (cfg == 0) cfg = 41;
        if (cfg == 41) return launch_cfg<7, 2, 2, 1, 1, 4, 32, 2>(ctx, p, s);   // 2-channel chunks: 4 spill 256 B/lane (13 weight float4 + 15 halo elements per thread)
        if (cfg == 42) return launch_cfg<7, 2, 2, 1, 1, 4, 32, 4>(ctx, p, s);
    } else if (a.ksize == 1 && a.stride == 2) {
        if (cfg == 0) cfg = p.Wout >= 24 ? 31 : (p.Wout >= 12 ? 32 : 33);
        switch (cfg) {
            case 31: return launch_cfg<1, 2, 2, 1, 1, 4, 32, 16>(ctx, p, s);
            case 32: return launch_cfg<1, 2, 2, 1, 1, 4, 16, 16>(ctx, p, s);
            case 33: return launch_cfg<1, 2, 1, 1, 2, 2, 8, 16>(ctx, p, s);
        }
    } else if (a.ksize == 1) {
        // 1x1: the image is a flat row of H*W pixels
        p.Hin = 1; p.Win = a.Hin * a.Win; p.Hc = 1; p.Wc = p.Win; p.Hout = 1; p.Wout = p.Win; p.ups = 0;
        SISIC_REQUIRE(!a.upsample, "conv2d: 1x1 with upsample");
        // tile_cfg 20: the lean pointwise kernel (conv_pointwise.hip) for the shapes it takes -- whole 128-pixel tiles and
        // 32-channel chunks -- unless SISIC_POINTWISE=0; the generic tilings below for everything else
        static const bool pw_on = [] { const char* e = std::getenv("SISIC_POINTWISE"); return !e || std::atoi(e) != 0; }();
        // tile_cfg 28: the same GEMM with fp32-equivalent products on the bf16 matrix pipe (conv_pointwise_bf3.hip) for whole
        // 64-pixel x 64-channel tiles -- unless SISIC_POINTWISE_BF16X3=0.  (Shape conditions only: an image's bits must not
        // depend on the batch it is in.)
        static const bool pwb_on = [] { const char* e = std::getenv("SISIC_POINTWISE_BF16X3"); return !e || std::atoi(e) != 0; }();
        if ((cfg == 0 && pwb_on && conv_pointwise_bf3_applicable(a)) || (cfg >= 28 && cfg <= 30) || cfg == 34 || cfg == 35) {
            if (slots_query) { *slots_query = conv_pointwise_stats_slots(a); return SISIC_OK; }
            return launch_conv_pointwise_bf3(ctx, a, s);
        }
        if ((cfg == 0 && pw_on && conv_pointwise_applicable(a)) || cfg == 20) {
            if (slots_query) { *slots_query = conv_pointwise_stats_slots(a); return SISIC_OK; }
            return launch_conv_pointwise(ctx, a, s);
        }
        if (cfg == 0) cfg = (p.Wout <= 64) ? 22 : (p.Wout <= 256 ? 25 : 24);   // measured (tools/conv_bench.py, B=64)
        switch (cfg) {
            case 21: return launch_cfg<1, 1, 2, 2, 1, 4, 256, 16>(ctx, p, s);
            case 22: return launch_cfg<1, 1, 1, 1, 2, 2, 64, 16>(ctx, p, s);
            case 23: return launch_cfg<1, 1, 2, 1, 1, 4, 128, 16>(ctx, p, s);
            case 24: return launch_cfg<1, 1, 1, 2, 2, 4, 256, 32, 4>(ctx, p, s);
            case 25: return launch_cfg<1, 1, 1, 1, 2, 4, 128, 32, 4>(ctx, p, s);
            case 26: return launch_cfg<1, 1, 1, 2, 2, 4, 256, 16, 4>(ctx, p, s);
            case 27: return launch_cfg<1, 1, 1, 1, 2, 4, 128, 16, 4>(ctx, p, s);
            // (four-wave tiles at three / six workgroups per CU were measured too -- profiles/r02/conv1x1_tilings_and_contraction.txt,
            //  cfg 28 / 29: within 2 % of 24 / slower -- and removed: occupancy is not what limits these launches)
        }
    } else if (a.stride == 1) {
        if (cfg == 0) cfg = p.Wout >= 24 ? 8 : (p.Wout >= 12 ? 9 : 16);   // measured (tools/conv_bench.py, B=64)
        switch (cfg) {
            case 1: return launch_cfg<3, 1, 2, 2, 1, 4, 64, 8>(ctx, p, s);
            case 2: return launch_cfg<3, 1, 2, 2, 1, 4, 32, 8>(ctx, p, s);
            case 3: return launch_cfg<3, 1, 2, 2, 1, 4, 16, 8>(ctx, p, s);
            case 4: return launch_cfg<3, 1, 1, 1, 2, 2, 8, 8>(ctx, p, s);
            case 16: return launch_cfg<3, 1, 1, 1, 2, 2, 8, 16, 2, 2>(ctx, p, s);   // 8 waves: 2 K groups x (2x2), 16-channel chunks
            case 17: return launch_cfg<3, 1, 1, 1, 2, 2, 8, 8, 2, 2>(ctx, p, s);
            case 5: return launch_cfg<3, 1, 2, 1, 1, 4, 16, 8>(ctx, p, s);
            case 6: return launch_cfg<3, 1, 2, 1, 1, 4, 64, 8>(ctx, p, s);
            case 7: return launch_cfg<3, 1, 2, 1, 1, 4, 32, 8>(ctx, p, s);
            case 8: return launch_cfg<3, 1, 1, 2, 2, 4, 32, 8, 4>(ctx, p, s);
            case 9: return launch_cfg<3, 1, 1, 2, 2, 4, 16, 8, 4>(ctx, p, s);
            case 10: return launch_cfg<3, 1, 1, 2, 2, 4, 32, 8, 2>(ctx, p, s);
            case 14: return launch_cfg<3, 1, 1, 1, 2, 4, 16, 8, 4>(ctx, p, s);
            case 15: return launch_cfg<3, 1, 1, 1, 2, 4, 8, 8, 4>(ctx, p, s);
        }
    } else {
        if (cfg == 0) cfg = p.Wout >= 24 ? 11 : (p.Wout >= 12 ? 12 : 18);   // measured (tools/conv_bench.py, B=64; 16->8: 66 -> 57 us)
        switch (cfg) {
            case 11: return launch_cfg<3, 2, 2, 1, 1, 4, 32, 8>(ctx, p, s);
            case 12: return launch_cfg<3, 2, 2, 1, 1, 4, 16, 8>(ctx, p, s);
            case 13: return launch_cfg<3, 2, 1, 1, 2, 2, 8, 8>(ctx, p, s);
            case 18: return launch_cfg<3, 2, 1, 1, 2, 2, 8, 16, 2, 2>(ctx, p, s);   // latency mode: 8x8 pixels, 8 waves = 2 K groups x (2x2)
            case 19: return launch_cfg<3, 2, 1, 1, 2, 2, 8, 8, 2, 2>(ctx, p, s);
        }
    }
    set_error("conv2d: tile_cfg %d invalid for ksize %d stride %d", cfg, a.ksize, a.stride);
    return SISIC_EINVAL;
}

}  // namespace sisic
