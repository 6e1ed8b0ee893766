// conv1x1.hip -- 1x1 stride-1 convolution (the UNet's shortcut, q/k/v and output projections) as a GEMM whose waves do
// not talk to each other.
//
// What the direct kernel's counters said about its flat 1x1 tiles (conv_mfma.hip, tile_cfg 24 / 25):
// a wave owns 32 channels x 64 pixels, so every k-step is one weight read + two pixel reads from LDS for two MFMAs --
// 1.5 ds_read per MFMA -- and on gfx950 an LDS-read stream adds ~70 % of its own issue time to the f32 matrix stream of
// its SIMD (tools/mfma_valu_probe).  That is the 0.73 marginal efficiency every tiling of that kernel converged to.
//
// Here a wave owns 32 consecutive pixels of one image x ALL channels of a 128- (or 64-) channel tile:
//   * weights never touch LDS: the wave loads its A fragments straight from global memory, from a second packing of the
//     matrix ([8-channel chunk][32-channel block][lane][4 channel pairs]: one 16-byte load per lane and row block is the
//     four k-steps of a chunk) -- the idea of conv_winograd_wide.inc.  Every wave of a channel tile reads the same
//     fragments; they come from L1 / L2 (the matrix is 128-512 KB);
//   * the pixel operand of a chunk is ONE 16-byte load per lane (8 channels x 32 pixels, 128 contiguous bytes per channel),
//     GroupNorm scale / shift (+SiLU) applied in registers, then transposed into the MFMA layout through a 1-KB LDS region
//     that only this wave touches: one ds_write_b128 + four ds_read_b32 per 16 (8) MFMAs = 0.25 (0.5) reads per MFMA;
//   * no barrier anywhere: LDS operations of one wave execute in order, and no other wave reads the region.
// A workgroup is four such waves (four neighbouring pixel blocks of the same channel tile) and exists only to be scheduled.
//
// Needs H*W % 32 == 0 (a wave's 32 pixels lie inside one image plane) and 16-byte aligned inputs; everything else stays
// on conv_mfma.hip.  Same summation order as the direct kernel with 8-channel chunks: channel pairs in ascending order.
//
// Algorithmic bytes: 4*B*(Cin + Cout)*HW (+ residual) + 4*Cin*Cout;  FLOPs 2*B*Cout*HW*Cin.  f32 MFMA bound.
#include "common.h"

namespace sisic {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Conv1Params {
    const float* in0;
    const float* in1;
    int c0, c1, B, HW;
    const float* w2;            // second packing (see conv1x1_pack_kernel)
    int n_co32;                 // 32-channel row blocks in the packing (Cout padded to 128)
    const float* bias;
    int Cout;
    const float* gn_scale;
    const float* gn_shift;
    const float* chan_bias;
    int chan_bias_stride;
    const float* residual;
    int relu;
    float* out;
    float* stats;               // optional [B][Cout][HW/32][4]
    int nchunks, n_blocks, blocks_per_image, n_px_wgs, n_co_tiles, nwg;
};

__device__ __forceinline__ float c1_half_wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v + __shfl_xor(v, 16);
}

constexpr int C1_WAVES = 4;       // waves (pixel blocks) per workgroup
constexpr int C1_CIC = 8;         // channels per chunk = four k-steps of v_mfma_f32_32x32x2_f32

// MT = 32-channel row blocks per wave (4: 128-channel tile, 2: 64-channel tile); PRO = 0 none, 1 GroupNorm, 2 GroupNorm+SiLU
template <int MT, int PRO>
__global__ void __launch_bounds__(64 * C1_WAVES, 4) conv1x1_kernel(const Conv1Params p) {
    __shared__ __attribute__((aligned(16))) float stage[C1_WAVES][2][C1_CIC * 32];     // per wave: two chunks of [8 ci][32 px]

    int work;
    {   // XCD-aware bijective remap (conv_mfma.hip): consecutive work items share an XCD's L2
        const int L = blockIdx.x, nwg = p.nwg;
        const int xcd = L & 7, slot = L >> 3, q = nwg >> 3, r = nwg & 7;
        work = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    // pixel groups fastest inside a channel tile: the waves resident on a CU read the same weight fragments at about the same time
    const int co_t = work / p.n_px_wgs;
    const int pg = work % p.n_px_wgs;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int blk = pg * C1_WAVES + wave;                  // 32-pixel block over all images
    if (blk >= p.n_blocks) return;                         // (no barrier in this kernel)
    const int b = blk / p.blocks_per_image;
    const int px0 = (blk % p.blocks_per_image) * 32;
    const int Cin = p.c0 + p.c1;
    const int co0 = co_t * MT * 32;

    // ---- pixel operand plan: lane = (channel of the chunk, four consecutive pixels)
    const int sci = lane >> 3, spx = (lane & 7) * 4;
    const float* const base0 = p.in0 + (size_t)b * p.c0 * p.HW + px0 + spx;
    const float* const base1 = p.c1 ? p.in1 + (size_t)b * p.c1 * p.HW + px0 + spx : base0;
    float* const my_stage = &stage[wave][0][0];
    const int st_off = sci * 32 + spx;                     // ds_write_b128 target inside a chunk buffer
    const int rd_off = half * 32 + l31;                    // k-step ks reads [2 ks + half][l31]

    struct XRegs {
        f32x4 v;
        float gsc, gsh;
        bool valid;
    };
    auto load_x = [&](int chunk, XRegs& x) {
        const int c = chunk * C1_CIC + sci;
        x.valid = c < Cin;
        const int cc = min(c, Cin - 1);
        const float* src = cc < p.c0 ? base0 + (size_t)cc * p.HW : base1 + (size_t)(cc - p.c0) * p.HW;
        x.v = *reinterpret_cast<const f32x4*>(src);
        if constexpr (PRO != 0) {
            x.gsc = p.gn_scale[(size_t)b * Cin + cc];
            x.gsh = p.gn_shift[(size_t)b * Cin + cc];
        }
    };
    auto stage_x = [&](int buf, const XRegs& x) {
        f32x4 v = x.v;
        if constexpr (PRO != 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = v[i] * x.gsc + x.gsh;
                if constexpr (PRO == 2) t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
                v[i] = t;
            }
        }
        if (!x.valid) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // channels past Cin: zero AFTER the prologue
        *reinterpret_cast<f32x4*>(my_stage + buf * (C1_CIC * 32) + st_off) = v;
    };
    auto read_b = [&](int buf, float (&bv)[4]) {
        const float* s = my_stage + buf * (C1_CIC * 32) + rd_off;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bv[ks] = s[ks * 64];
    };
    auto load_a = [&](int chunk, f32x4 (&a)[MT]) {
        const float* base = p.w2 + (((size_t)chunk * p.n_co32 + (size_t)co_t * MT) * 64 + lane) * 4;
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(base + (size_t)m * 256);
    };

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;

    // ---- pipeline (per wave, no barriers).  Entering iteration c: registers hold the fragments A(c), the pixel operand
    // B(c) in MFMA layout, and the raw pixels X(c+1), X(c+2) still in flight.  The iteration stages X(c+1) through LDS into
    // B(c+1), requests X(c+3) and A(c+1), and issues the MFMAs of chunk c.
    const int n = p.nchunks;
    f32x4 a_cur[MT], a_nxt[MT];
    float b_cur[4], b_nxt[4];
    XRegs x1, x2;
    {
        XRegs x0;
        load_x(0, x0);
        load_a(0, a_cur);
        load_x(min(1, n - 1), x1);
        load_x(min(2, n - 1), x2);
        stage_x(0, x0);
        read_b(0, b_cur);
    }
    auto step = [&](int c, f32x4 (&ac)[MT], f32x4 (&an)[MT], float (&bc)[4], float (&bn)[4], XRegs& xa, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;       // buffer holding chunk c
        load_a(min(c + 1, n - 1), an);
        stage_x(PAR ^ 1, xa);                               // X(c+1) -> LDS
        load_x(min(c + 3, n - 1), xa);                      // (re-uses the registers just staged)
        read_b(PAR ^ 1, bn);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[m][ks], bc[ks], acc[m], 0, 0, 0);
    };
    int c = 0;
    for (; c + 1 < n; c += 2) {
        step(c, a_cur, a_nxt, b_cur, b_nxt, x1, std::integral_constant<int, 0>{});
        step(c + 1, a_nxt, a_cur, b_nxt, b_cur, x2, std::integral_constant<int, 1>{});
    }
    if (c < n) step(c, a_cur, a_nxt, b_cur, b_nxt, x1, std::integral_constant<int, 0>{});

    // ---- epilogue: accumulator (m, r) of a lane is channel m*32 + 8*(r/4) + 4*half + r%4 at pixel px0 + l31: each
    // half-wave stores 128 contiguous bytes per channel.  Loads clamped and unconditional, stores predicated.
    const size_t plane0 = (size_t)b * p.Cout * p.HW + px0 + l31;
    const int slots = p.blocks_per_image, slot = blk % p.blocks_per_image;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float add[16], res[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = min(co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, p.Cout - 1);
            float t = 0.0f;
            if (p.bias) t += p.bias[co];
            if (p.chan_bias) t += p.chan_bias[(size_t)b * p.chan_bias_stride + co];
            add[r] = t;
            res[r] = p.residual ? p.residual[plane0 + (size_t)co * p.HW] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = acc[m][r] + add[r] + res[r];
            if (p.relu) v = fmaxf(v, 0.0f);
            if (co < p.Cout) p.out[plane0 + (size_t)co * p.HW] = v;
            acc[m][r] = v;
        }
        if (p.stats) {      // (count, sum, centred M2) of this wave's 32 pixels of each channel: one slot per pixel block
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float s1 = c1_half_wave_sum(acc[m][r]);
                const float d = acc[m][r] - s1 * (1.0f / 32.0f);
                const float q = c1_half_wave_sum(d * d);
                if (l31 == 0 && co < p.Cout)
                    reinterpret_cast<float4*>(p.stats)[((size_t)b * p.Cout + co) * slots + slot] = make_float4(32.0f, s1, q, 0.0f);
            }
        }
    }
}

// second packing of a 1x1 weight matrix (OIHW with H = W = 1):
//   [chunk = ci / 8][row block = co / 32][lane = (ci & 1) * 32 + co % 32][channel pair = (ci % 8) / 2], zero padded
__global__ void conv1x1_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int cin_pad8, int cout_pad128,
                                    float* __restrict__ out) {
    const size_t total = (size_t)cin_pad8 * cout_pad128;
    const int n_co32 = cout_pad128 >> 5;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout_pad128), ci = (int)(i / cout_pad128);
        const int chunk = ci >> 3, cp = (ci & 7) >> 1, hf = ci & 1;
        const size_t o = ((((size_t)chunk * n_co32 + (co >> 5)) * 64) + hf * 32 + (co & 31)) * 4 + cp;
        out[o] = (co < Cout && ci < Cin) ? w[(size_t)co * Cin + ci] : 0.0f;
    }
}

int64_t conv1x1_second_numel(int Cout, int Cin) { return (int64_t)round_up(Cin, C1_CIC) * round_up(Cout, 128); }

int launch_conv1x1_pack(sisic_ctx*, const float* w, int Cout, int Cin, float* out, hipStream_t s) {
    const int cin_pad8 = round_up(Cin, C1_CIC), cout_pad128 = round_up(Cout, 128);
    const size_t total = (size_t)cin_pad8 * cout_pad128;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(conv1x1_pack_kernel, dim3(blocks), dim3(256), 0, s, w, Cout, Cin, cin_pad8, cout_pad128, out);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// whether launch_conv1x1 can take these arguments (the rest of the checks are the caller's: dispatch_conv2d)
bool conv1x1_applicable(const sisic_conv_args& a) {
    const int HW = a.Hin * a.Win;
    const uintptr_t al = reinterpret_cast<uintptr_t>(a.in0) | reinterpret_cast<uintptr_t>(a.in1);
    return a.ksize == 1 && a.stride == 1 && !a.upsample && HW % 32 == 0 && (al & 15) == 0 &&
           (double)a.B * std::max(a.c0, a.c1) * HW < 2147483648.0;
}
int conv1x1_stats_slots(const sisic_conv_args& a) { return a.Hin * a.Win / 32; }

template <int MT, int PRO>
static int launch_c1(const Conv1Params& p, hipStream_t s) {
    hipLaunchKernelGGL((conv1x1_kernel<MT, PRO>), dim3(p.nwg), dim3(64 * C1_WAVES), 0, s, p);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// w2: the second packing (a.w_packed + first numel).  wide: 128-channel tiles, else 64.
int launch_conv1x1(sisic_ctx* ctx, const sisic_conv_args& a, const float* w2, bool wide, hipStream_t s) {
    SISIC_REQUIRE(conv1x1_applicable(a), "conv2d(1x1 stream form): needs ksize 1, stride 1, H*W %% 32 == 0 and 16-byte aligned inputs");
    Conv1Params p{};
    p.in0 = a.in0; p.in1 = a.in1; p.c0 = a.c0; p.c1 = a.c1; p.B = a.B; p.HW = a.Hin * a.Win;
    p.w2 = w2; p.n_co32 = round_up(a.Cout, 128) / 32;
    p.bias = a.bias; p.Cout = a.Cout;
    p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift;
    p.chan_bias = a.chan_bias; p.chan_bias_stride = a.chan_bias_stride; p.residual = a.residual; p.relu = a.relu;
    p.out = a.out; p.stats = a.stats_out;
    p.nchunks = cdiv(a.c0 + a.c1, C1_CIC);
    p.blocks_per_image = p.HW / 32;
    const int64_t n_blocks = (int64_t)a.B * p.blocks_per_image;
    const int mt = wide ? 4 : 2;
    p.n_co_tiles = cdiv(a.Cout, mt * 32);
    const int64_t n_px_wgs = (n_blocks + C1_WAVES - 1) / C1_WAVES;
    const int64_t nwg = n_px_wgs * p.n_co_tiles;
    SISIC_REQUIRE(nwg > 0 && nwg < (int64_t(1) << 31), "conv2d(1x1 stream form): grid too large");
    p.n_blocks = (int)n_blocks; p.n_px_wgs = (int)n_px_wgs; p.nwg = (int)nwg;
    const int pro = (a.gn_scale == nullptr) ? 0 : (a.gn_silu ? 2 : 1);
    if (wide) {
        if (pro == 2) return launch_c1<4, 2>(p, s);
        if (pro == 1) return launch_c1<4, 1>(p, s);
        return launch_c1<4, 0>(p, s);
    }
    if (pro == 2) return launch_c1<2, 2>(p, s);
    if (pro == 1) return launch_c1<2, 1>(p, s);
    return launch_c1<2, 0>(p, s);
}

}  // namespace sisic
