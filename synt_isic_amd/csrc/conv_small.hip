// conv_small.hip -- 3x3 stride-1 convolution with at most 4 output channels (the UNet's conv_out, 64 -> 3).
//
// On the MFMA kernel a 3-channel output pads to a 64-row tile (95 % of the matrix work wasted; 160 us at
// B=64, 64x64).  With so few outputs the layer is bandwidth/LDS-bound (14 MFLOP per image against 1 MB of
// input), so it runs on the vector ALU instead.  A thread owns FOUR horizontally adjacent output pixels of a
// 32x32 tile: the six input values of a row it needs are one ds_read_b128 + one ds_read_b64 and are reused by
// the 3 taps x 4 pixels x 4 filters that touch them (a pixel per thread re-read every input nine times and was
// LDS-bound at 87 us); the <=4 filters of a tap are one broadcast ds_read_b128.  The input halo tile is staged
// through LDS exactly like conv_mfma.hip (GroupNorm+SiLU prologue on the way in, zero padding after it,
// two-source concat), two channels per chunk so that a staging thread moves ten elements.
// Same packed weight layout as conv_mfma.hip ([Cin_pad][9][Cout_pad]): the first 4 floats of each row.
//
// Algorithmic bytes: 4*B*(Cin + Cout)*H*W (+ weights); HBM-bound.
#include <atomic>
#include <type_traits>

#include "common.h"

namespace sisic {

// Tile = 32 x CS_TH output pixels.  CS_TH = 32 (256 threads) when the launch has workgroups to spare; 8 (one wave) when it
// does not (a single 128x128 image is 16 tiles of 32x32).  A pixel's sum runs over the channels in the same order either
// way, so the choice never changes a bit of the result.
constexpr int CS_TW = 32, CS_PX = 4, CS_CIC = 2;
constexpr int CS_IW = CS_TW + 2;
constexpr int CS_IWP = 36;                                       // LDS row stride: float4-aligned rows
template <int CS_TH>
struct CSGeom;
template <int CS_TH, int KS>
constexpr size_t cs_lds_bytes();
template <int CS_TH>
struct CSGeom {
    static constexpr int THR = (CS_TW / CS_PX) * CS_TH;          // 256 / 64 threads
    static constexpr int IH = CS_TH + 2;
    static constexpr int TPC = THR / CS_CIC;                     // threads staging one channel
    static constexpr int EPT = (IH * CS_IWP + TPC - 1) / TPC;    // 10 / 12
    static constexpr int CHS = EPT * TPC;                        // floats per channel (padded)
    static_assert(CS_IWP >= CS_IW + 2 && CS_IWP % 4 == 0 && CHS % 4 == 0, "aligned rows");
    static_assert(CS_CIC * 9 <= THR, "one thread per filter row");
};

template <int CS_TH, int KS>
constexpr size_t cs_lds_bytes() {
    constexpr size_t in = (size_t)KS * 2 * CS_CIC * CSGeom<CS_TH>::CHS, red = (size_t)(KS - 1) * CS_PX * 4 * CSGeom<CS_TH>::THR;
    return sizeof(float) * (in > red ? in : red);
}

struct ConvSmallParams {
    const float* in0;
    const float* in1;
    int c0, c1, B, H, W;
    const float* w;
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
    int tiles_x, tiles_y, nchunks;
};

// KS (round 4): the input channels split over KS thread groups of one workgroup (each walks 1 / KS of the chunks of the SAME
// pixel tile, its own LDS buffers), their partial sums added through LDS in the fixed order ((g0 + g1) + g2) + g3.  At batch 64
// the 32-row tiling is 256 workgroups of four waves -- ONE wave per SIMD, 32 chunks of [load, stage, barrier, multiply] each:
// latency-bound at 48 us for 67 MB of input; with four groups the chip has four waves per SIMD.  Another summation order, so
// the choice is by the layer's shape alone (launcher), never by the batch.
template <int CS_TH, int KS>
__global__ void __launch_bounds__(CSGeom<CS_TH>::THR * KS, KS == 1 ? 3 : 4) conv3x3_smallcout_kernel(const ConvSmallParams p) {
    constexpr int CS_THR = CSGeom<CS_TH>::THR, CS_IH = CSGeom<CS_TH>::IH, CS_TPC = CSGeom<CS_TH>::TPC, CS_EPT = CSGeom<CS_TH>::EPT,
                  CS_CHS = CSGeom<CS_TH>::CHS;
    constexpr int IN_FLOATS = 2 * CS_CIC * CS_CHS, RED_FLOATS = (KS - 1) * CS_PX * 4 * CS_THR;
    extern __shared__ __attribute__((aligned(16))) float lds_all[];      // max(KS * IN_FLOATS, RED_FLOATS) floats (launcher)
    __shared__ __attribute__((aligned(16))) float w_all[KS][2][CS_CIC * 9 * 4];
    const int grp = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / CS_THR);     // (CS_THR is a multiple of 64)
    float (*in_lds)[CS_CIC * CS_CHS] = reinterpret_cast<float (*)[CS_CIC * CS_CHS]>(lds_all + grp * IN_FLOATS);
    float (*w_lds)[CS_CIC * 9 * 4] = w_all[grp];

    int tile = blockIdx.x;
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int ty = tile % p.tiles_y;
    const int b = tile / p.tiles_y;
    const int oy0 = ty * CS_TH, ox0 = tx * CS_TW;
    const int tid = threadIdx.x - grp * CS_THR;              // the thread's place in its group
    const int sci = tid / CS_TPC, sl = tid % CS_TPC;
    const int HW = p.H * p.W, Cin = p.c0 + p.c1;
    const int prologue = (p.gn_scale == nullptr) ? 0 : (p.gn_silu ? 2 : 1);
    const int n_mine = p.nchunks / KS, c_first = grp * n_mine;       // this group's chunks (launcher: nchunks % KS == 0)

    int goff[CS_EPT];
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < CS_EPT; ++i) {
        const int e = sl + i * CS_TPC;                       // position in the padded [CS_IH][CS_IWP] halo tile
        const int yy = e / CS_IWP, xx = e % CS_IWP;
        const int gy = oy0 - 1 + yy, gx = ox0 - 1 + xx;
        const bool v = yy < CS_IH && xx < CS_IW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        goff[i] = v ? gy * p.W + gx : 0;
        vmask |= (v ? 1u : 0u) << i;
    }

    float rin[CS_EPT];
    float rw[4];
    float gsc = 1.0f, gsh = 0.0f;
    bool cval = false;

    auto load_chunk = [&](int chunk) {
        const int c = chunk * CS_CIC + sci;
        cval = c < Cin;
        const int cc = min(c, Cin - 1);
        const float* src = (cc < p.c0) ? p.in0 + ((size_t)b * p.c0 + cc) * HW : p.in1 + ((size_t)b * p.c1 + (cc - p.c0)) * HW;
#pragma unroll
        for (int i = 0; i < CS_EPT; ++i) rin[i] = src[goff[i]];
        if (prologue) {
            gsc = p.gn_scale[(size_t)b * Cin + cc];
            gsh = p.gn_shift[(size_t)b * Cin + cc];
        }
        // 18 rows (ci, tap) x 4 filters per chunk: threads 0..17 fetch one float4 each
        const int row = min(tid, CS_CIC * 9 - 1);
        const float4 t = *reinterpret_cast<const float4*>(p.w + ((size_t)chunk * CS_CIC * 9 + row) * p.cout_pad);
        rw[0] = t.x; rw[1] = t.y; rw[2] = t.z; rw[3] = t.w;
    };
    auto store_chunk = [&](int buf) {
        float* dst = &in_lds[buf][sci * CS_CHS + sl];
        const unsigned m = cval ? vmask : 0u;
#pragma unroll
        for (int i = 0; i < CS_EPT; ++i) {
            float v = rin[i];
            if (prologue) v = v * gsc + gsh;
            if (prologue == 2) v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
            dst[i * CS_TPC] = ((m >> i) & 1u) ? v : 0.0f;
        }
        if (tid < CS_CIC * 9) *reinterpret_cast<float4*>(&w_lds[buf][tid * 4]) = make_float4(rw[0], rw[1], rw[2], rw[3]);
    };

    const int py = tid / (CS_TW / CS_PX), px0 = (tid % (CS_TW / CS_PX)) * CS_PX;
    float acc[CS_PX][4];
#pragma unroll
    for (int q = 0; q < CS_PX; ++q)
#pragma unroll
        for (int co = 0; co < 4; ++co) acc[q][co] = 0.0f;
    load_chunk(c_first);
    store_chunk(0);
    __syncthreads();
    for (int k = 0; k < n_mine; ++k) {
        const int chunk = c_first + k;
        const int buf = k & 1;
        const bool more = k + 1 < n_mine;
        if (more) load_chunk(chunk + 1);
        const float* I = &in_lds[buf][py * CS_IWP + px0];
        const float* Wt = &w_lds[buf][0];
#pragma unroll 1
        for (int ci = 0; ci < CS_CIC; ++ci) {             // not unrolled: 18 taps' filters in flight would cost 72 registers
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* row = I + ci * CS_CHS + ky * CS_IWP;
                const float4 r4 = *reinterpret_cast<const float4*>(row);           // halo columns px0 .. px0+3
                const float2 r2 = *reinterpret_cast<const float2*>(row + 4);       //              px0+4, px0+5
                const float in[6] = {r4.x, r4.y, r4.z, r4.w, r2.x, r2.y};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 w4 = *reinterpret_cast<const float4*>(&Wt[(ci * 9 + ky * 3 + kx) * 4]);
#pragma unroll
                    for (int q = 0; q < CS_PX; ++q) {
                        const float a = in[q + kx];
                        acc[q][0] += a * w4.x; acc[q][1] += a * w4.y; acc[q][2] += a * w4.z; acc[q][3] += a * w4.w;
                    }
                }
            }
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }
    if constexpr (KS > 1) {
        // partial sums of groups 1 .. KS-1 through LDS (over the staging buffers: the loop's last barrier is behind every read)
        if (grp > 0) {
#pragma unroll
            for (int q = 0; q < CS_PX; ++q)
#pragma unroll
                for (int co = 0; co < 4; ++co) lds_all[((grp - 1) * CS_PX * 4 + q * 4 + co) * CS_THR + tid] = acc[q][co];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g = 1; g < KS; ++g)
#pragma unroll
            for (int q = 0; q < CS_PX; ++q)
#pragma unroll
                for (int co = 0; co < 4; ++co) acc[q][co] += lds_all[((g - 1) * CS_PX * 4 + q * 4 + co) * CS_THR + tid];
    }

    const int oy = oy0 + py, ox = ox0 + px0;
    if (oy < p.H) {
        const bool vec = (p.W & 3) == 0;           // then rows and planes are float4-aligned and ox % 4 == 0
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            if (co < p.Cout) {
                const size_t base = ((size_t)b * p.Cout + co) * HW + (size_t)oy * p.W;
                float add = 0.0f;
                if (p.bias) add += p.bias[co];
                if (p.chan_bias) add += p.chan_bias[(size_t)b * p.chan_bias_stride + co];
                float v[CS_PX];
#pragma unroll
                for (int q = 0; q < CS_PX; ++q) {
                    v[q] = acc[q][co] + add;
                    if (p.residual && ox + q < p.W) v[q] += p.residual[base + ox + q];
                    if (p.relu) v[q] = fmaxf(v[q], 0.0f);
                }
                if (vec) {
                    if (ox < p.W) *reinterpret_cast<float4*>(p.out + base + ox) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int q = 0; q < CS_PX; ++q)
                        if (ox + q < p.W) p.out[base + ox + q] = v[q];
                }
            }
        }
    }
}

// Used by launch_conv2d for ksize 3, stride 1, no upsample, Cout <= 4.
int launch_conv_smallcout(sisic_ctx* ctx, const sisic_conv_args& a, hipStream_t s) {
    ConvSmallParams p{};
    p.in0 = a.in0; p.in1 = a.in1; p.c0 = a.c0; p.c1 = a.c1; p.B = a.B; p.H = a.Hin; p.W = a.Win;
    p.w = a.w_packed; p.cout_pad = conv_cout_pad(a.Cout); p.bias = a.bias; p.Cout = a.Cout;
    p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.gn_silu = a.gn_silu;
    p.chan_bias = a.chan_bias; p.chan_bias_stride = a.chan_bias_stride; p.residual = a.residual; p.relu = a.relu;
    p.out = a.out;
    p.tiles_x = cdiv(a.Win, CS_TW);
    p.nchunks = cdiv(a.c0 + a.c1, CS_CIC);
    // tile_cfg 50 / 51 force the 32- / 8-row tile (tools/conv_bench.py); otherwise 32 rows from one workgroup per CU up
    const bool spare = a.tile_cfg == 50 || (a.tile_cfg != 51 && (int64_t)a.B * p.tiles_x * cdiv(a.Hin, 32) >= 256);
    p.tiles_y = cdiv(a.Hin, spare ? 32 : 8);
    const int64_t nwg = (int64_t)a.B * p.tiles_x * p.tiles_y;
    SISIC_REQUIRE(nwg > 0 && nwg < (int64_t(1) << 31), "conv2d(small): grid too large");
    // four channel groups per workgroup from 32 input channels up (a rule of the layer's shape: the groups' sums are added in
    // their own order); tile_cfg 52 forces one group
    const bool ksplit = a.tile_cfg != 52 && p.nchunks % 4 == 0 && p.nchunks >= 16;
    auto go = [&](auto th_tag, auto ks_tag) -> int {
        constexpr int TH = decltype(th_tag)::value, KSV = decltype(ks_tag)::value;
        auto kern = conv3x3_smallcout_kernel<TH, KSV>;
        static std::atomic<uint64_t> opt{0};
        SISIC_TRY(ensure_dynamic_lds(ctx, reinterpret_cast<const void*>(kern), (int)(cs_lds_bytes<TH, KSV>()), opt));
        constexpr size_t lds = cs_lds_bytes<TH, KSV>();
        constexpr unsigned thr = (unsigned)(CSGeom<TH>::THR * KSV);
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(thr), lds, s, p);
        return SISIC_OK;
    };
    using I = std::integral_constant<int, 1>;
    if (ksplit) {
        if (spare) SISIC_TRY(go(std::integral_constant<int, 32>{}, std::integral_constant<int, 4>{}));
        else SISIC_TRY(go(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{}));
    } else {
        if (spare) SISIC_TRY(go(std::integral_constant<int, 32>{}, I{}));
        else SISIC_TRY(go(std::integral_constant<int, 8>{}, I{}));
    }
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

}  // namespace sisic
