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
#include "common.h"

namespace sisic {

// Tile = 32 x CS_TH output pixels.  CS_TH = 32 (256 threads) when the launch has workgroups to spare; 8 (one wave) when it
// does not (a single 128x128 image is 16 tiles of 32x32).  A pixel's sum runs over the channels in the same order either
// way, so the choice never changes a bit of the result.
constexpr int CS_TW = 32, CS_PX = 4, CS_CIC = 2;
constexpr int CS_IW = CS_TW + 2;
constexpr int CS_IWP = 36;                                       // LDS row stride: float4-aligned rows
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

template <int CS_TH>
__global__ void __launch_bounds__(CSGeom<CS_TH>::THR, 3) conv3x3_smallcout_kernel(const ConvSmallParams p) {
    constexpr int CS_THR = CSGeom<CS_TH>::THR, CS_IH = CSGeom<CS_TH>::IH, CS_TPC = CSGeom<CS_TH>::TPC, CS_EPT = CSGeom<CS_TH>::EPT,
                  CS_CHS = CSGeom<CS_TH>::CHS;
    __shared__ __attribute__((aligned(16))) float in_lds[2][CS_CIC * CS_CHS];
    __shared__ __attribute__((aligned(16))) float w_lds[2][CS_CIC * 9 * 4];

    int tile = blockIdx.x;
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int ty = tile % p.tiles_y;
    const int b = tile / p.tiles_y;
    const int oy0 = ty * CS_TH, ox0 = tx * CS_TW;
    const int tid = threadIdx.x;
    const int sci = tid / CS_TPC, sl = tid % CS_TPC;
    const int HW = p.H * p.W, Cin = p.c0 + p.c1;
    const int prologue = (p.gn_scale == nullptr) ? 0 : (p.gn_silu ? 2 : 1);

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
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const int buf = chunk & 1;
        const bool more = chunk + 1 < p.nchunks;
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
    if (spare) hipLaunchKernelGGL(conv3x3_smallcout_kernel<32>, dim3((unsigned)nwg), dim3(CSGeom<32>::THR), 0, s, p);
    else hipLaunchKernelGGL(conv3x3_smallcout_kernel<8>, dim3((unsigned)nwg), dim3(CSGeom<8>::THR), 0, s, p);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

}  // namespace sisic
