// conv_small.hip -- 3x3 stride-1 convolution with at most 4 output channels (the UNet's conv_out, 64 -> 3).
//
// On the MFMA kernel a 3-channel output pads to a 64-row tile (95 % of the matrix work wasted; 160 us at
// B=64, 64x64).  With so few outputs the layer is bandwidth/LDS-bound (14 MFLOP per image against 1 MB of
// input), so it runs on the vector ALU instead: one thread per output pixel of a 16x16 tile, the input halo
// tile staged through LDS exactly like conv_mfma.hip (GroupNorm+SiLU prologue on the way in, zero padding
// after it, two-source concat), the <=4 filters of a tap read as one broadcast ds_read_b128.
// Same packed weight layout as conv_mfma.hip ([Cin_pad][9][Cout_pad]): the first 4 floats of each row.
//
// Algorithmic bytes: 4*B*(Cin + Cout)*H*W (+ weights); HBM-bound.
#include "common.h"

namespace sisic {

constexpr int CS_TW = 16, CS_TH = 16, CS_CIC = 8, CS_THR = 256;
constexpr int CS_IW = CS_TW + 2, CS_IH = CS_TH + 2;
constexpr int CS_TPC = CS_THR / CS_CIC;                          // 32 threads stage one channel
constexpr int CS_EPT = (CS_IH * CS_IW + CS_TPC - 1) / CS_TPC;    // 11
constexpr int CS_CHS = CS_EPT * CS_TPC;                          // 352 floats per channel (padded)

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

__global__ void __launch_bounds__(CS_THR) conv3x3_smallcout_kernel(const ConvSmallParams p) {
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
        const int e = sl + i * CS_TPC;
        const int gy = oy0 - 1 + e / CS_IW, gx = ox0 - 1 + e % CS_IW;
        const bool v = e < CS_IH * CS_IW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
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
        // 72 rows (ci, tap) x 4 filters per chunk: threads 0..71 fetch one float4 each
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

    const int py = tid / CS_TW, px = tid % CS_TW;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const int buf = chunk & 1;
        const bool more = chunk + 1 < p.nchunks;
        if (more) load_chunk(chunk + 1);
        const float* I = &in_lds[buf][py * CS_IW + px];
        const float* Wt = &w_lds[buf][0];
#pragma unroll
        for (int ci = 0; ci < CS_CIC; ++ci) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float a = I[ci * CS_CHS + ky * CS_IW + kx];
                    const float4 w4 = *reinterpret_cast<const float4*>(&Wt[(ci * 9 + ky * 3 + kx) * 4]);
                    acc[0] += a * w4.x; acc[1] += a * w4.y; acc[2] += a * w4.z; acc[3] += a * w4.w;
                }
            }
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }

    const int oy = oy0 + py, ox = ox0 + px;
    if (oy < p.H && ox < p.W) {
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            if (co < p.Cout) {
                const size_t idx = ((size_t)b * p.Cout + co) * HW + (size_t)oy * p.W + ox;
                float v = acc[co];
                if (p.bias) v += p.bias[co];
                if (p.chan_bias) v += p.chan_bias[(size_t)b * p.chan_bias_stride + co];
                if (p.residual) v += p.residual[idx];
                if (p.relu) v = fmaxf(v, 0.0f);
                p.out[idx] = v;
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
    p.tiles_x = cdiv(a.Win, CS_TW); p.tiles_y = cdiv(a.Hin, CS_TH);
    p.nchunks = cdiv(a.c0 + a.c1, CS_CIC);
    const int64_t nwg = (int64_t)a.B * p.tiles_x * p.tiles_y;
    SISIC_REQUIRE(nwg > 0 && nwg < (int64_t(1) << 31), "conv2d(small): grid too large");
    hipLaunchKernelGGL(conv3x3_smallcout_kernel, dim3((unsigned)nwg), dim3(CS_THR), 0, s, p);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

}  // namespace sisic
