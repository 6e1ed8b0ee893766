// conv_winograd.hip -- 3x3 stride-1 convolution as Winograd F(2x2,3x3) on the gfx950 f32 MFMA pipe.
//
// The direct implicit-GEMM kernel (conv_mfma.hip) is bound by the fp32 matrix rate (157 TFLOP/s): 49 of the
// UNet's 52 conv3x3 are stride 1, and for those the minimal-filtering form needs 16 multiplies per 2x2
// output tile and channel pair instead of 36 -- 2.25x fewer MFMA FLOPs at unchanged fp32 arithmetic:
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A          (Lavin & Gray; all transform coefficients are 0, +-1, +-1/2)
// Per workgroup (512 threads = 8 waves): 64 output channels x 64 tiles (2x2 output pixels each).  The input
// channels are walked in chunks of 8:
//   stage   the halo tile [8][(2TY+2)x(2TX+2)] -> LDS (GroupNorm+SiLU prologue, zero padding after it, concat,
//           nearest-2x upsampling: the same load path as conv_mfma.hip), the transformed filters U[8][16][64]
//           (pre-computed in float64 at load time) -> LDS
//   xform   wave w = channel w of the chunk, lane = tile: V = B^T d B, 32 adds, -> LDS V[8][16][64]
//   mfma    wave w owns the two transform positions xi = 2w, 2w+1:  M_xi[co,tile] += U_xi[co,ci] V_xi[ci,tile]
//           with v_mfma_f32_32x32x2_f32, 2x2 register tiles per xi (128 accumulator registers)
// U and V are double-buffered so staging/transform of chunk c+1 overlaps the MFMAs of chunk c.  At the end
// the 16 M_xi planes are exchanged through LDS in four 16-channel rounds and every thread applies
// Y = A^T M A for its (channel, tile) pairs, adds bias / time embedding / residual, and stores 2x2 pixels.
//
// Numerics: fp32 throughout; filter transform in float64 rounded once.  Max error vs a float64 conv is a
// small multiple of the direct kernel's (tests: same 2e-5 * max(1,|ref|) bound).
#include "common.h"

namespace sisic {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoParams {
    const float* in0;
    const float* in1;
    int c0, c1;
    int B, Hin, Win;    // source tensors
    int Hc, Wc;         // conv extent (= output extent), after the optional 2x upsample
    int ups;
    const float* u;     // packed transformed filters [Cin_pad][16][cout_pad]
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
    int groups_x, groups_y, groups_b, n_co_tiles, nwg, nchunks;
};

constexpr int WG_THREADS = 512;
constexpr int W_CIC = 8;            // input channels per chunk
constexpr int W_TILES = 64;         // 2x2 output tiles per workgroup
constexpr int W_CO = 64;            // output channels per workgroup
constexpr int W_SLAB = W_CIC * 16 * 64;   // floats of one U or V buffer (32 KiB)

__device__ __forceinline__ float wsilu(float v) { return __fdividef(v, 1.0f + __expf(-v)); }

template <int NIMG, int TY, int TX>
struct WinoGeom {
    static_assert(NIMG * TY * TX == W_TILES, "64 tiles per workgroup");
    static constexpr int HH = 2 * TY + 2, HWD = 2 * TX + 2;   // halo extent per image
    static constexpr int HPI = HH * HWD;
    static constexpr int HEL = NIMG * HPI;                    // halo elements per channel
    static constexpr int TPC = WG_THREADS / W_CIC;            // 64 threads stage one channel
    static constexpr int EPT = (HEL + TPC - 1) / TPC;
    static constexpr int CHS = EPT * TPC;                     // padded channel stride in LDS
    static constexpr size_t LDS_BYTES = (size_t)(4 * W_SLAB + W_CIC * CHS) * sizeof(float);
    static_assert(EPT <= 32, "valid mask is 32 bits");
};

template <int NIMG, int TY, int TX>
__global__ void __launch_bounds__(WG_THREADS, 2) conv_winograd_kernel(const WinoParams p) {
    using G = WinoGeom<NIMG, TY, TX>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const U_lds = smem;                   // [2][W_SLAB]   ([ci][xi][co])
    float* const V_lds = smem + 2 * W_SLAB;      // [2][W_SLAB]   ([ci][xi][tile])
    float* const H_lds = smem + 4 * W_SLAB;      // [W_CIC][CHS]  halo tile of the chunk being transformed
    float* const M_lds = smem;                   // epilogue: [16 xi][16 co][64 tiles] over the U/V buffers

    int work;
    {   // XCD-aware bijective remap (see conv_mfma.hip)
        const int L = blockIdx.x, nwg = p.nwg;
        const int xcd = L & 7, slot = L >> 3, q = nwg >> 3, r = nwg & 7;
        work = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int co_t = work % p.n_co_tiles;
    int grp = work / p.n_co_tiles;
    const int gx = grp % p.groups_x;
    grp /= p.groups_x;
    const int gy = grp % p.groups_y;
    const int gb = grp / p.groups_y;
    const int oy0 = gy * 2 * TY, ox0 = gx * 2 * TX, b0 = gb * NIMG;
    const int co0 = co_t * W_CO;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int HWin = p.Hin * p.Win, Cin = p.c0 + p.c1;
    const int prologue = (p.gn_scale == nullptr) ? 0 : (p.gn_silu ? 2 : 1);

    // ---- halo staging plan: wave w stages channel w of the chunk, lanes stride over its halo elements
    const int sci = wave, sl = lane;
    int goff[G::EPT];
    int gimg[G::EPT];
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < G::EPT; ++i) {
        const int e = sl + i * G::TPC;
        const int img = e / G::HPI, r = e % G::HPI;
        const int yy = r / G::HWD, xx = r % G::HWD;
        const int y = oy0 - 1 + yy, x = ox0 - 1 + xx;
        const bool v = e < G::HEL && (b0 + img) < p.B && y >= 0 && y < p.Hc && x >= 0 && x < p.Wc;
        goff[i] = v ? ((y >> p.ups) * p.Win + (x >> p.ups)) : 0;
        gimg[i] = v ? img : 0;
        vmask |= (v ? 1u : 0u) << i;
    }

    // ---- transform plan: wave = channel, lane = tile
    int xf_base;
    {
        const int t = lane;
        const int img = t / (TY * TX), ty = (t / TX) % TY, tx = t % TX;
        xf_base = sci * G::CHS + img * G::HPI + (2 * ty) * G::HWD + 2 * tx;
    }

    // ---- MFMA operand bases: wave w owns xi = 2w, 2w+1
    const int ab_base = half * 16 * 64 + (2 * wave) * 64 + l31;

    f32x16 acc[2][2][2];     // [xi][co tile][tile tile]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][m][n][r] = 0.0f;

    float rin[G::EPT];
    float rw[16];
    float gsc = 1.0f, gsh = 0.0f;
    bool cval = false;

    auto load_chunk = [&](int chunk) {     // global -> registers (all loads unconditional at clamped addresses)
        const int c = chunk * W_CIC + sci;
        cval = c < Cin;
        const int cc = min(c, Cin - 1);
        const bool first = cc < p.c0;
        const float* src = first ? p.in0 + ((size_t)b0 * p.c0 + cc) * HWin : p.in1 + ((size_t)b0 * p.c1 + (cc - p.c0)) * HWin;
        const size_t img_stride = (size_t)(first ? p.c0 : p.c1) * HWin;
#pragma unroll
        for (int i = 0; i < G::EPT; ++i) rin[i] = src[(NIMG > 1 ? gimg[i] * img_stride : 0) + goff[i]];
        if (prologue) {
            gsc = p.gn_scale[(size_t)b0 * Cin + cc];      // NIMG > 1: per-image scale/shift applied in stage_halo
            gsh = p.gn_shift[(size_t)b0 * Cin + cc];
        }
        const float* usrc = p.u + (size_t)chunk * W_CIC * 16 * p.cout_pad + co0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + i * WG_THREADS;            // 2048 float4 per slab
            const int rr = f >> 4, c4 = f & 15;
            const float4 t = *reinterpret_cast<const float4*>(usrc + (size_t)rr * p.cout_pad + c4 * 4);
            rw[4 * i + 0] = t.x; rw[4 * i + 1] = t.y; rw[4 * i + 2] = t.z; rw[4 * i + 3] = t.w;
        }
    };

    auto stage = [&](int chunk, int buf) {      // registers -> H_lds (prologue applied) and U_lds[buf]
        float* dst = H_lds + sci * G::CHS + sl;
        const unsigned m = cval ? vmask : 0u;
        const int cc = min(chunk * W_CIC + sci, Cin - 1);
#pragma unroll
        for (int i = 0; i < G::EPT; ++i) {
            float v = rin[i];
            if (prologue) {
                float sc = gsc, sh = gsh;
                if (NIMG > 1) {     // scale/shift differ per image of the group
                    const int bi = min(b0 + gimg[i], p.B - 1);
                    sc = p.gn_scale[(size_t)bi * Cin + cc];
                    sh = p.gn_shift[(size_t)bi * Cin + cc];
                }
                v = v * sc + sh;
                if (prologue == 2) v = wsilu(v);
            }
            dst[i * G::TPC] = ((m >> i) & 1u) ? v : 0.0f;
        }
        float* udst = U_lds + buf * W_SLAB + tid * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(udst + i * WG_THREADS * 4) =
                make_float4(rw[4 * i + 0], rw[4 * i + 1], rw[4 * i + 2], rw[4 * i + 3]);
    };

    auto transform = [&](int buf) {             // H_lds -> V_lds[buf]:  V = B^T d B for this wave's channel
        const float* hp = H_lds + xf_base;
        float d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 a = *reinterpret_cast<const float2*>(hp + i * G::HWD);
            const float2 b = *reinterpret_cast<const float2*>(hp + i * G::HWD + 2);
            d[i][0] = a.x; d[i][1] = a.y; d[i][2] = b.x; d[i][3] = b.y;
        }
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j];
            t[1][j] = d[1][j] + d[2][j];
            t[2][j] = d[2][j] - d[1][j];
            t[3][j] = d[1][j] - d[3][j];
        }
        float* vp = V_lds + buf * W_SLAB + sci * 16 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vp[(4 * i + 0) * 64] = t[i][0] - t[i][2];
            vp[(4 * i + 1) * 64] = t[i][1] + t[i][2];
            vp[(4 * i + 2) * 64] = t[i][2] - t[i][1];
            vp[(4 * i + 3) * 64] = t[i][1] - t[i][3];
        }
    };

    auto mfma_part = [&](int buf, int cp0, int cp1) {
        const float* A = U_lds + buf * W_SLAB + ab_base;
        const float* Bm = V_lds + buf * W_SLAB + ab_base;
#pragma unroll
        for (int cp = cp0; cp < cp1; ++cp) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                float a[2], b[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) a[m] = A[(2 * cp) * 16 * 64 + x * 64 + m * 32];
#pragma unroll
                for (int n = 0; n < 2; ++n) b[n] = Bm[(2 * cp) * 16 * 64 + x * 64 + n * 32];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[x][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[x][m][n], 0, 0, 0);
            }
        }
    };

    // ---- software pipeline over the channel chunks
    load_chunk(0);
    stage(0, 0);
    __syncthreads();
    transform(0);
    if (p.nchunks > 1) load_chunk(1);
    __syncthreads();
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const int buf = chunk & 1;
        const bool more = chunk + 1 < p.nchunks;
        if (more) stage(chunk + 1, buf ^ 1);              // H_lds <- chunk+1, U_lds[buf^1] <- chunk+1
        mfma_part(buf, 0, 2);
        __syncthreads();                                  // H_lds visible
        if (more) transform(buf ^ 1);                     // V_lds[buf^1] <- chunk+1
        if (chunk + 2 < p.nchunks) load_chunk(chunk + 2); // prefetch two chunks ahead into registers
        mfma_part(buf, 2, 4);
        __syncthreads();                                  // U/V[buf^1] visible, U/V[buf] free
    }

    // ---- output transform: four rounds of 16 output channels through LDS
    const size_t HWout = (size_t)p.Hc * p.Wc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mt = q >> 1, rbase = 8 * (q & 1);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = rbase + rr;
                    const int row16 = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;     // row within the 16-channel block
                    M_lds[((2 * wave + x) * 16 + row16) * 64 + n * 32 + l31] = acc[x][mt][n][r];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int pi = tid + k * WG_THREADS;       // 1024 (channel, tile) pairs
            const int co16 = pi >> 6, t = pi & 63;
            float m[4][4];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) m[xi >> 2][xi & 3] = M_lds[(xi * 16 + co16) * 64 + t];
            float s[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[0][j] = m[0][j] + m[1][j] + m[2][j];
                s[1][j] = m[1][j] - m[2][j] - m[3][j];
            }
            float y[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                y[i][0] = s[i][0] + s[i][1] + s[i][2];
                y[i][1] = s[i][1] - s[i][2] - s[i][3];
            }
            const int co = co0 + mt * 32 + 16 * (q & 1) + co16;
            const int img = t / (TY * TX), ty = (t / TX) % TY, tx = t % TX;
            const int b = b0 + img;
            const int oy = oy0 + 2 * ty, ox = ox0 + 2 * tx;
            const bool ok = co < p.Cout && b < p.B;
            const int coc = min(co, p.Cout - 1), bc = min(b, p.B - 1);
            float add = 0.0f;
            if (p.bias) add += p.bias[coc];
            if (p.chan_bias) add += p.chan_bias[(size_t)bc * p.chan_bias_stride + coc];
            const size_t plane = ((size_t)bc * p.Cout + coc) * HWout;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int yy = oy + i, xx = ox + j;
                    const bool in = ok && yy < p.Hc && xx < p.Wc;
                    const size_t idx = plane + (size_t)min(yy, p.Hc - 1) * p.Wc + min(xx, p.Wc - 1);
                    float v = y[i][j] + add;
                    if (p.residual) v += p.residual[idx];
                    if (p.relu) v = fmaxf(v, 0.0f);
                    if (in) p.out[idx] = v;
                }
            }
        }
        __syncthreads();
    }
}

// U = G g G^T per (co, ci), float64 arithmetic, written as [Cin_pad][16][cout_pad] (zero padded)
__global__ void winograd_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int cin_pad, int cout_pad,
                                     float* __restrict__ out) {
    const size_t total = (size_t)cin_pad * cout_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
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
}

int launch_winograd_pack(sisic_ctx*, const float* w, int Cout, int Cin, float* packed, hipStream_t s) {
    const int cin_pad = round_up(Cin, W_CIC), cout_pad = conv_cout_pad(Cout);
    const size_t total = (size_t)cin_pad * cout_pad;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(winograd_pack_kernel, dim3(blocks), dim3(256), 0, s, w, Cout, Cin, cin_pad, cout_pad, packed);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

int64_t winograd_packed_numel(int Cout, int Cin) { return (int64_t)round_up(Cin, W_CIC) * 16 * conv_cout_pad(Cout); }

template <int NIMG, int TY, int TX>
static int launch_wino(sisic_ctx* ctx, WinoParams& p, hipStream_t s) {
    using G = WinoGeom<NIMG, TY, TX>;
    p.groups_x = cdiv(p.Wc, 2 * TX);
    p.groups_y = cdiv(p.Hc, 2 * TY);
    p.groups_b = cdiv(p.B, NIMG);
    p.n_co_tiles = p.cout_pad / W_CO;
    p.nchunks = cdiv(p.c0 + p.c1, W_CIC);
    const int64_t nwg = (int64_t)p.groups_x * p.groups_y * p.groups_b * p.n_co_tiles;
    SISIC_REQUIRE(nwg > 0 && nwg < (int64_t(1) << 31), "conv2d(winograd): grid too large");
    p.nwg = (int)nwg;
    auto kern = conv_winograd_kernel<NIMG, TY, TX>;
    static bool attr_set = false;
    if (!attr_set) {
        SISIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)G::LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.nwg), dim3(WG_THREADS), G::LDS_BYTES, s, p);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// tile_cfg 60: 1 image x 8x8 tiles (16x16 output pixels);  61: 4 images x 4x4 tiles (8x8 output pixels each)
int launch_conv_winograd(sisic_ctx* ctx, const sisic_conv_args& a, const float* u_packed, int cfg, hipStream_t s) {
    WinoParams p{};
    p.in0 = a.in0; p.in1 = a.in1; p.c0 = a.c0; p.c1 = a.c1;
    p.B = a.B; p.Hin = a.Hin; p.Win = a.Win;
    p.ups = a.upsample ? 1 : 0;
    p.Hc = a.Hin << p.ups; p.Wc = a.Win << p.ups;
    p.u = u_packed; p.cout_pad = conv_cout_pad(a.Cout);
    p.bias = a.bias; p.Cout = a.Cout;
    p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.gn_silu = a.gn_silu;
    p.chan_bias = a.chan_bias; p.chan_bias_stride = a.chan_bias_stride; p.residual = a.residual; p.relu = a.relu;
    p.out = a.out;
    if (cfg == 61) return launch_wino<4, 4, 4>(ctx, p, s);
    return launch_wino<1, 8, 8>(ctx, p, s);
}

}  // namespace sisic
