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
#include <type_traits>

#include "common.h"
#include "gn_merge.h"
#include "pack_device.h"

// Timing-only ablation builds for tools/conv_bench.py (results are wrong): bit 0 skips the MFMAs, bit 1 the
// output transform, bit 2 the halo staging + input transform.  Never defined in the shipped library.
// WINO_TIMING (diagnostic build only, never in the shipped library): wave 0 of every workgroup stamps s_memtime at its
// phase boundaries into p.stats ([nwg][8] uint64) instead of the GroupNorm partials -- tools/wino_phases.py reads them.
#ifndef WINO_TIMING
#define WINO_TIMING 0
#endif
#define WINO_STAMP_SLOTS 10      // uint64 per workgroup: 8 phase stamps + HW_ID + XCC_ID (third form)
#if WINO_TIMING
#define WINO_STAMP(slot)                                                                                   \
    do {                                                                                                   \
        unsigned long long t_;                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
        if (tid == 0) reinterpret_cast<unsigned long long*>(p.stats)[(size_t)blockIdx.x * WINO_STAMP_SLOTS + (slot)] = t_; \
    } while (0)
#else
#define WINO_STAMP(slot) do { } while (0)
#endif
#ifndef WINO_ABLATE
#define WINO_ABLATE 0
#endif

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
    const float* uw;    // the same filters in the wide form's packing (conv_winograd_wide.inc), behind the first
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
    float* stats;       // optional [B][Cout][groups_y*groups_x][4]: per-workgroup (count, sum, centred M2, 0) of the stored values
    float* part;        // K-split: raw partial outputs [ksplit][B][Cout][Hc][Wc], summed by splitk_reduce_kernel
    int ksplit;         // workgroups per output tile, each covering nchunks/ksplit channel chunks (1 = no split)
    int groups_x, groups_y, groups_b, n_co_tiles, nwg, nchunks;
    int stagger;        // 1: half of the waves run MFMA-first, the other half stage-first (see the channel loop)
    // sisic_conv_args.fin_*: where a workgroup holds a whole image of its channels (conv_winograd_bf3.inc, one tile per image) it
    // finalizes the following GroupNorm's groups of eight channels itself (fin_gamma == nullptr: not requested / not possible)
    const float* fin_gamma;
    const float* fin_beta;
    float* fin_scale;
    float* fin_shift;
    float* fin_mean_rstd;
    int fin_groups;
    float fin_eps;
};

constexpr int W_CIC = 8;            // input channels per chunk
constexpr int W_TILES = 64;         // 2x2 output tiles per workgroup
constexpr int W_CO = 64;            // output channels per workgroup
constexpr int W_SLAB = W_CIC * 16 * 64;   // floats of one U or V buffer (32 KiB)

// Makes a wave-uniform pointer provably uniform for hipcc (two v_readfirstlane), so that the loads through it use the
// scalar-base form  global_load v, v_offset32, s[base:base+1]  instead of per-lane 64-bit address arithmetic.
// The result is typed as a GLOBAL (address space 1) pointer: rebuilt from two integers the pointer would otherwise be
// generic and every load through it a flat_load, which counts in lgkmcnt as well as vmcnt -- the s_waitcnt lgkmcnt(0) in
// front of each MFMA's LDS operands then also waited for the global prefetches issued just before (found in the ISA of
// round 1's kernel: 1501 flat_load against 553 global_load).
typedef const __attribute__((address_space(1))) char* gcptr_t;
typedef const __attribute__((address_space(1))) float* gfptr_t;
typedef float v4f_t __attribute__((ext_vector_type(4)));     // plain vector types: HIP's float4 / float2 are classes whose
typedef float v2f_t __attribute__((ext_vector_type(2)));     // constructors do not bind to address-space-1 references
typedef const __attribute__((address_space(1))) v4f_t* gf4ptr_t;
typedef const __attribute__((address_space(1))) v2f_t* gf2ptr_t;
typedef __attribute__((address_space(1))) float* gfwptr_t;
typedef __attribute__((address_space(1))) v2f_t* gf2wptr_t;
__device__ __forceinline__ gcptr_t uniform_ptr(const void* ptr) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (gcptr_t)(((unsigned long long)hi << 32) | lo);
}

// x * sigmoid(x) with the hardware reciprocal (v_rcp_f32, 1 ulp): __fdividef expands to the 10-instruction IEEE
// division sequence, and every VALU instruction in this loop displaces matrix work (tools/mfma_valu_probe.hip).
__device__ __forceinline__ float wsilu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// Sum over each aligned run of 16 lanes (DPP butterflies: quad swaps, then the two row mirrors); every lane of
// the run receives the total.  Fixed order, so the GroupNorm partials are bit-reproducible.
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}
__device__ __forceinline__ float wave64_sum(float v) {
    const int r = __float_as_int(row16_sum(v));
    return (__int_as_float(__builtin_amdgcn_readlane(r, 0)) + __int_as_float(__builtin_amdgcn_readlane(r, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(r, 32)) + __int_as_float(__builtin_amdgcn_readlane(r, 48)));
}

// NW = waves per workgroup: 8 (two transform positions xi per wave, 128 accumulator registers, 2 waves/SIMD)
//                       or 16 (one xi per wave, 64 accumulator registers, 4 waves/SIMD).
template <int NIMG, int TY, int TX, int NW, bool UPS = false>
struct WinoGeom {
    static_assert(NIMG * TY * TX == W_TILES, "64 tiles per workgroup");
    static_assert(NW == 8 || NW == 16, "8 or 16 waves");
    static constexpr int THREADS = 64 * NW;
    static constexpr int XPW = 16 / NW;                       // transform positions per wave
    // halo extent per image; the upsample form stages the LOW-resolution halo (each 4x4 patch is a 3x3 patch there)
    static constexpr int HH = UPS ? TY + 2 : 2 * TY + 2, HWD = UPS ? TX + 2 : 2 * TX + 2;
    static constexpr int HPI = HH * HWD;
    static constexpr int HEL = NIMG * HPI;                    // halo elements per channel
    static constexpr int TPC = THREADS / W_CIC;               // threads staging one channel
    // Several images per workgroup: when a channel's staging threads cover one image's halo (HPI <= TPC), element i of
    // a thread IS image i -- the image index is then a compile-time constant (scalar GroupNorm operands, no per-lane
    // image bookkeeping).  Otherwise the HEL elements are dealt out flat.
    static constexpr bool PER_IMAGE = NIMG > 1 && HPI <= TPC;
    static constexpr int EPT = PER_IMAGE ? NIMG : (HEL + TPC - 1) / TPC;
    static constexpr int CHS = PER_IMAGE ? ((HEL + 1) / 2) * 2 : EPT * TPC;   // (padded) channel stride in LDS
    static constexpr int HBUF = W_CIC * CHS;
    static constexpr int UPT = (W_SLAB / 4) / THREADS;        // float4 of the U slab per thread
    static constexpr size_t LDS_BYTES = (size_t)(4 * W_SLAB + 2 * HBUF) * sizeof(float);
    static_assert(EPT <= 32, "valid mask is 32 bits");
    static_assert(CHS % 2 == 0 && HPI % 2 == 0, "float2 transform reads need even strides");
    static_assert(!UPS || NIMG == 1, "upsample form: one image per workgroup");
};

// UPS (nearest-2x upsampled input, 16 waves): every 4x4 input patch of an even-aligned tile has rows (a, b, b, c) and
// columns likewise, so B^T d B vanishes on transform row 2 and column 2 -- 7 of the 16 positions are identically zero.
// Only the other nine are multiplied: eight waves own one position each, four waves share the ninth (one 32x32 tile
// each), four waves only stage; with waves dealt round-robin to the SIMDs every SIMD carries 9 of the 36 tile products
// per k-step instead of 16.  Zero positions are neither loaded (filters), written (V) nor read back (output transform).
constexpr unsigned WINO_UPS_ZERO = 0x4F44u;        // bit xi set: position xi = 4*row + col has row == 2 or col == 2
constexpr unsigned WINO_UPS_XI = 0xDC754310u;      // nibble w: position of full wave w (0,1,3,4,5,7,12,13); 15 is shared

template <int NIMG, int TY, int TX, int PRO, int NW, bool UPS = false>   // PRO: 0 = no prologue, 1 = GroupNorm apply, 2 = + SiLU
__global__ void __launch_bounds__(64 * NW, NW / 4) conv_winograd_kernel(const WinoParams p) {
    using G = WinoGeom<NIMG, TY, TX, NW, UPS>;
    static_assert(!UPS || (NW == 16 && NIMG == 1), "the upsample form is built for one position per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const U_lds = smem;                   // [2][W_SLAB]   ([ci][xi][co])
    float* const V_lds = smem + 2 * W_SLAB;      // [2][W_SLAB]   ([ci][xi][tile])
    float* const H_lds = smem + 4 * W_SLAB;      // [2][HBUF]     staged halo tiles (prologue applied)
    float* const M_lds = smem;                   // epilogue: [16 xi][16 co][64 tiles] over the U/V buffers

    int work;
    {   // XCD-aware bijective remap (see conv_mfma.hip)
        const int L = blockIdx.x, nwg = p.nwg;
        const int xcd = L & 7, slot = L >> 3, q = nwg >> 3, r = nwg & 7;
        work = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int co_t = work % p.n_co_tiles;
    int grp = work / p.n_co_tiles;
    const int ks = grp % p.ksplit;          // which share of the input channels (K-split)
    grp /= p.ksplit;
    const int gx = grp % p.groups_x;
    grp /= p.groups_x;
    const int gy = grp % p.groups_y;
    const int gb = grp / p.groups_y;
    const int oy0 = gy * 2 * TY, ox0 = gx * 2 * TX, b0 = gb * NIMG;
    const int co0 = co_t * W_CO;
    const int n = p.nchunks / p.ksplit;     // chunks of this workgroup (the host makes ksplit divide nchunks)
    const int cbase = ks * n;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int HWin = p.Hin * p.Win, Cin = p.c0 + p.c1;
    WINO_STAMP(0);

    // ---- halo staging plan: TPC threads per channel of the chunk.  The staged channel is wave-uniform
    // (TPC is a multiple of 64), which lets every chunk-dependent address term live in SGPRs: global loads take the
    // form  scalar base pointer (per chunk) + 32-bit per-thread byte offset (loop-invariant), i.e. the saddr
    // encoding of global_load, and the channel loop carries no vector address arithmetic at all.
    // (__builtin_amdgcn_raw_buffer_load_b128 is NOT used: hipcc 7.2 lowers it to a single buffer_load_dword.)
    // (only the staged channel is made provably uniform: a scalar `wave` makes hipcc unswitch the loops and spill)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int sci = (G::TPC == 64) ? wave_u : (wave_u >> 1);
    const int sl = tid % G::TPC;
    unsigned goff[G::EPT];        // BYTE offset of the element inside its (image, channel) plane
    int gimg[G::EPT];             // image of the element (a constant i under PER_IMAGE)
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < G::EPT; ++i) {
        const int e = G::PER_IMAGE ? i * G::HPI + sl : sl + i * G::TPC;
        const int img = G::PER_IMAGE ? i : e / G::HPI, r = G::PER_IMAGE ? sl : e % G::HPI;
        const int yy = r / G::HWD, xx = r % G::HWD;
        bool v = (G::PER_IMAGE ? sl < G::HPI : e < G::HEL) && (b0 + img) < p.B;
        if constexpr (UPS) {      // low-resolution coordinates; hi-res padding rows -1 / Hc are low-res rows -1 / Hin
            const int y = (oy0 >> 1) - 1 + yy, x = (ox0 >> 1) - 1 + xx;
            v = v && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
            goff[i] = v ? 4u * (unsigned)(y * p.Win + x) : 0u;
        } else {
            const int y = oy0 - 1 + yy, x = ox0 - 1 + xx;
            v = v && y >= 0 && y < p.Hc && x >= 0 && x < p.Wc;
            goff[i] = v ? 4u * (unsigned)((y >> p.ups) * p.Win + (x >> p.ups)) : 0u;
        }
        gimg[i] = v ? img : 0;       // images past the batch read (and discard) image b0: never an address outside the tensor
        vmask |= (v ? 1u : 0u) << i;
    }
    const bool stage_lane = !G::PER_IMAGE || sl < G::HPI;      // PER_IMAGE: threads past the halo stage nothing
    unsigned uoff[G::UPT];        // byte offset of this thread's float4 inside a filter slab
#pragma unroll
    for (int i = 0; i < G::UPT; ++i) {
        const int f = tid + i * G::THREADS;                // 2048 float4 per slab
        uoff[i] = 4u * (unsigned)((f >> 4) * p.cout_pad + (f & 15) * 4);
    }

    // ---- transform plan: lane = tile; NW=8: wave = channel (whole 4x4 patch); NW=16: wave = (channel, row half)
    const int xci = (NW == 8) ? wave : (wave >> 1);
    int xf_base;
    {
        const int t = lane;
        const int img = t / (TY * TX), ty = (t / TX) % TY, tx = t % TX;
        xf_base = UPS ? xci * G::CHS + ty * G::HWD + tx : xci * G::CHS + img * G::HPI + (2 * ty) * G::HWD + 2 * tx;
    }

    // ---- MFMA operand bases: wave w owns xi = w*XPW .. w*XPW + XPW-1
    // UPS roles (wave-uniform): 0 = owns position xi_w, 1 = one 32x32 tile (qm, qn) of position 15, 2 = staging only
    const int role = !UPS ? 0 : (wave_u < 8 ? 0 : (wave_u < 12 ? 1 : 2));
    const int xi_w = !UPS ? wave * G::XPW : (wave_u < 8 ? (int)((WINO_UPS_XI >> (4 * wave_u)) & 15u) : 15);
    const int qm = (wave_u >> 1) & 1, qn = wave_u & 1;
    const int ab_base = half * 16 * 64 + xi_w * 64 + l31;

    f32x16 acc[G::XPW][2][2];     // [xi][co tile][tile tile]
#pragma unroll
    for (int x = 0; x < G::XPW; ++x)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][m][n][r] = 0.0f;

    struct HaloRegs {             // one chunk's halo elements in flight (global -> registers -> H_lds) and their prologue
        float v[G::EPT];
        float gsc = 1.0f, gsh = 0.0f;
        float gsc_i[G::EPT], gsh_i[G::EPT];      // PER_IMAGE: one GroupNorm pair per image (uniform -> scalar loads)
        bool cval = false;
        int hcc = 0;              // channel of the data held in v
    };
    struct FilterRegs { float w[G::UPT * 4]; };   // filter slab in flight
    HaloRegs hr;
    FilterRegs fr;

    // All loads are unconditional at clamped, always-valid addresses and masked afterwards (see conv_mfma.hip).
    auto load_halo = [&](int chunk, HaloRegs& h) {
        const int c = (cbase + chunk) * W_CIC + sci;       // scalar
        h.cval = c < Cin;
        const int hcc = h.hcc = min(c, Cin - 1);
        const bool first = hcc < p.c0;
        const int csrc = first ? p.c0 : p.c1;
        const gcptr_t plane = uniform_ptr(
            first ? p.in0 + ((size_t)b0 * p.c0 + hcc) * HWin : p.in1 + ((size_t)b0 * p.c1 + (hcc - p.c0)) * HWin);
        const unsigned img_stride = 4u * (unsigned)(csrc * HWin);
#pragma unroll
        for (int i = 0; i < G::EPT; ++i) {
            const unsigned vo = (NIMG > 1) ? (unsigned)gimg[i] * img_stride + goff[i] : goff[i];
            h.v[i] = *(gfptr_t)(plane + vo);
        }
        if constexpr (PRO != 0) {
            h.gsc = p.gn_scale[b0 * Cin + hcc];            // uniform index: scalar loads.  NIMG > 1: see stage_halo
            h.gsh = p.gn_shift[b0 * Cin + hcc];
            if constexpr (G::PER_IMAGE) {
#pragma unroll
                for (int i = 0; i < G::EPT; ++i) {
                    const int bi = min(b0 + i, p.B - 1);
                    h.gsc_i[i] = p.gn_scale[bi * Cin + hcc];
                    h.gsh_i[i] = p.gn_shift[bi * Cin + hcc];
                }
            }
        }
    };
    // a thread's float4 of the filter slab always belongs to position (tid >> 4) & 15
    const bool u_active = !UPS || !((WINO_UPS_ZERO >> ((tid >> 4) & 15)) & 1u);
    auto load_u = [&](int chunk, FilterRegs& f) {
        if (!u_active) return;
        const gcptr_t slab = uniform_ptr(p.u + (size_t)(cbase + chunk) * W_CIC * 16 * p.cout_pad + co0);
#pragma unroll
        for (int i = 0; i < G::UPT; ++i) {
            const v4f_t t = *(gf4ptr_t)(slab + uoff[i]);
            f.w[4 * i + 0] = t.x; f.w[4 * i + 1] = t.y; f.w[4 * i + 2] = t.z; f.w[4 * i + 3] = t.w;
        }
    };
    auto stage_halo = [&](int hbuf, const HaloRegs& h) {   // registers -> H_lds[hbuf], prologue applied, padding zeroed after it
        float* dst = H_lds + hbuf * G::HBUF + sci * G::CHS + sl;
        const unsigned m = h.cval ? vmask : 0u;
#pragma unroll
        for (int i = 0; i < G::EPT; ++i) {
            float v = h.v[i];
            if constexpr (PRO != 0) {
                float sc = h.gsc, sh = h.gsh;
                if constexpr (G::PER_IMAGE) {
                    sc = h.gsc_i[i];
                    sh = h.gsh_i[i];
                } else if constexpr (NIMG > 1) {
                    const int bi = min(b0 + gimg[i], p.B - 1);
                    sc = p.gn_scale[(size_t)bi * Cin + h.hcc];
                    sh = p.gn_shift[(size_t)bi * Cin + h.hcc];
                }
                v = v * sc + sh;
                if constexpr (PRO == 2) v = wsilu(v);
            }
            if (stage_lane) dst[i * (G::PER_IMAGE ? G::HPI : G::TPC)] = ((m >> i) & 1u) ? v : 0.0f;
        }
    };
    auto stage_u = [&](int buf, const FilterRegs& f) {
        if (!u_active) return;
        float* udst = U_lds + buf * W_SLAB + tid * 4;
#pragma unroll
        for (int i = 0; i < G::UPT; ++i)
            *reinterpret_cast<float4*>(udst + i * G::THREADS * 4) =
                make_float4(f.w[4 * i + 0], f.w[4 * i + 1], f.w[4 * i + 2], f.w[4 * i + 3]);
    };
    auto transform = [&](int hbuf, int vbuf, auto xrh_tag) {   // H_lds[hbuf] -> V_lds[vbuf]:  V = B^T d B
        constexpr int xrh = decltype(xrh_tag)::value;      // row half of this wave (NW = 16), fixed per loop instance
        const float* hp = H_lds + hbuf * G::HBUF + xf_base;
        float* vp = V_lds + vbuf * W_SLAB + xci * 16 * 64 + lane;
        if constexpr (UPS) {
            // 3x3 low-resolution patch L; the 4x4 hi-res patch has rows (L0, L1, L1, L2) and columns likewise, so
            // B^T d B reduces to  rows: L0-L1, 2 L1, (0), L1-L2  and the same three combinations along the columns.
            // This wave: transform rows 0 and 1 (xrh = 0, from L0, L1) or row 3 (xrh = 1, from L1, L2).
            float l[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) l[i][j] = hp[(i + xrh) * G::HWD + j];
            float d[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) d[j] = l[0][j] - l[1][j];              // rows 0 (xrh = 0) / 3 (xrh = 1)
            const int r0 = xrh ? 3 : 0;
            vp[(4 * r0 + 0) * 64] = d[0] - d[1];
            vp[(4 * r0 + 1) * 64] = 2.0f * d[1];
            vp[(4 * r0 + 3) * 64] = d[1] - d[2];
            if (!xrh) {                                                        // row 1 = 2 L1
                vp[(4 * 1 + 0) * 64] = 2.0f * (l[1][0] - l[1][1]);
                vp[(4 * 1 + 1) * 64] = 4.0f * l[1][1];
                vp[(4 * 1 + 3) * 64] = 2.0f * (l[1][1] - l[1][2]);
            }
        } else if constexpr (NW == 8) {
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
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vp[(4 * i + 0) * 64] = t[i][0] - t[i][2];
                vp[(4 * i + 1) * 64] = t[i][1] + t[i][2];
                vp[(4 * i + 2) * 64] = t[i][2] - t[i][1];
                vp[(4 * i + 3) * 64] = t[i][1] - t[i][3];
            }
        } else {
            // this wave produces V rows 2*xrh and 2*xrh+1: it needs d rows xrh .. xrh+2
            float d[3][4];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float2 a = *reinterpret_cast<const float2*>(hp + (i + xrh) * G::HWD);
                const float2 b = *reinterpret_cast<const float2*>(hp + (i + xrh) * G::HWD + 2);
                d[i][0] = a.x; d[i][1] = a.y; d[i][2] = b.x; d[i][3] = b.y;
            }
            float t[2][4];
            // xrh = 0: rows (d0,d1,d2): t0 = d0 - d2, t1 = d1 + d2;   xrh = 1: rows (d1,d2,d3): t2 = d2 - d1, t3 = d1 - d3
            if (xrh) {          // wave-uniform: a branch, not 8 selects
#pragma unroll
                for (int j = 0; j < 4; ++j) { t[0][j] = d[1][j] - d[0][j]; t[1][j] = d[0][j] - d[2][j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 2 * xrh + i;
                if (UPS && row == 2) continue;                 // identically zero (wave-uniform test)
                vp[(4 * row + 0) * 64] = t[i][0] - t[i][2];
                vp[(4 * row + 1) * 64] = t[i][1] + t[i][2];
                if (!UPS) vp[(4 * row + 2) * 64] = t[i][2] - t[i][1];
                vp[(4 * row + 3) * 64] = t[i][1] - t[i][3];
            }
        }
    };
    auto mfma_chunk = [&](int buf, auto role_tag) {
        constexpr int ROLE = decltype(role_tag)::value;
        const float* A = U_lds + buf * W_SLAB + ab_base;
        const float* Bm = V_lds + buf * W_SLAB + ab_base;
        if constexpr (ROLE == 2) return;
        if constexpr (ROLE == 1) {                             // one tile of the shared position, kept in acc[0][0][0]
#pragma unroll
            for (int cp = 0; cp < W_CIC / 2; ++cp)
                acc[0][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(2 * cp) * 16 * 64 + qm * 32],
                                                                    Bm[(2 * cp) * 16 * 64 + qn * 32], acc[0][0][0], 0, 0, 0);
            return;
        }
#pragma unroll
        for (int cp = 0; cp < W_CIC / 2; ++cp) {
#pragma unroll
            for (int x = 0; x < G::XPW; ++x) {
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

    // ---- software pipeline over the channel chunks, ONE barrier per chunk.  Entering iteration c:
    //   U/V[c&1] hold chunk c;  H[(c+1)&1] holds the staged halo of chunk c+1;  registers hold halo(c+2) and U(c+1).
    // Prologue: the loads of chunks 0 AND 1 are issued together (one exposed memory latency instead of two; the
    // accumulators are not live yet, so the second register set is free here).
    WINO_STAMP(1);               // index plans done
    {
        HaloRegs h1;
        FilterRegs f1;
        load_halo(0, hr);
        load_u(0, fr);
        if (n > 1) {
            load_halo(1, h1);
            load_u(1, f1);
        }
        stage_halo(0, hr);
        stage_u(0, fr);
        if (n > 2) load_halo(2, hr);
        __syncthreads();
        WINO_STAMP(2);           // first chunk loaded and staged
        if (NW == 16 && (wave & 1)) transform(0, 0, std::integral_constant<int, 1>{});
        else transform(0, 0, std::integral_constant<int, 0>{});
        if (n > 1) {
            stage_halo(1, h1);
            fr = f1;
        }
        __syncthreads();
    }
    WINO_STAMP(3);               // prologue done

    // Phase stagger: all waves of the workgroup run the same program and meet at one barrier per chunk, so left
    // alone they would all be in their staging phase (VALU/LDS) at the same time and all in their MFMA phase at the
    // same time -- the matrix pipe then idles during staging and the time is the SUM of the two (measured).  Waves
    // 0-3 (and 8-11) therefore issue the chunk's MFMAs FIRST and stage afterwards, waves 4-7 (and 12-15) stage
    // first: every SIMD hosts one wave of each kind (waves are dealt to SIMDs round-robin), so its matrix pipe
    // always has a wave feeding it while the partner wave does the vector/LDS work.  Everything inside one
    // barrier interval touches disjoint buffers, so the order within the interval is free.
    const bool mfma_first = p.stagger && (((wave >> 2) & 1) == 0);
    // The loop is instantiated per wave role (matrix-first or staging-first order, transform row half, upsample role)
    // and the role is tested ONCE, outside: a role test inside the loop costs exec-mask bookkeeping and ~20 register
    // moves per chunk (or, with scalar conditions, makes hipcc unswitch the loop and spill the accumulators).
    // Every instance executes the same barriers.
    auto channel_loop = [&](auto role_tag, auto first_tag, auto xrh_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        // One chunk of the steady state; PAR = c & 1 is a compile-time constant (the loop is unrolled by two), so every
        // LDS buffer offset is an instruction immediate instead of a per-use address addition.
        auto body = [&](const int c, auto par_tag) {
            constexpr int PAR = decltype(par_tag)::value;
            if constexpr (FIRST) {
#if !(WINO_ABLATE & 1)
                mfma_chunk(PAR, role_tag);
#endif
                transform(PAR ^ 1, PAR ^ 1, xrh_tag);
                stage_u(PAR ^ 1, fr);
                stage_halo(PAR, hr);
                load_halo(c + 3, hr);
                load_u(c + 2, fr);
            } else {
#if !(WINO_ABLATE & 4)
                transform(PAR ^ 1, PAR ^ 1, xrh_tag);
#endif
                stage_u(PAR ^ 1, fr);
#if !(WINO_ABLATE & 4)
                stage_halo(PAR, hr);                 // halo(c+2) -> H[c&1] (its previous content, halo(c), was consumed last iteration)
#endif
                load_halo(c + 3, hr);
                load_u(c + 2, fr);
#if !(WINO_ABLATE & 1)
                mfma_chunk(PAR, role_tag);
#endif
            }
            __syncthreads();
        };
        int c = 0;
        for (; c + 4 < n; c += 2) {              // steady state, two chunks per trip: one basic block
            body(c, std::integral_constant<int, 0>{});
            body(c + 1, std::integral_constant<int, 1>{});
        }
        if (c + 3 < n) {                         // odd leftover of the steady state
            body(c, std::integral_constant<int, 0>{});
            ++c;
        }
        for (; c < n; ++c) {                     // last (up to) three chunks: same order, guarded
            if (c + 1 < n) {
                transform((c + 1) & 1, (c + 1) & 1, xrh_tag);
                stage_u((c + 1) & 1, fr);
            }
            if (c + 2 < n) {
                stage_halo(c & 1, hr);
                load_u(c + 2, fr);
            }
#if !(WINO_ABLATE & 1)
            mfma_chunk(c & 1, role_tag);
#endif
            __syncthreads();
        }
    };
    auto by_order = [&](auto role_tag, auto xrh_tag) {
        if (mfma_first) channel_loop(role_tag, std::true_type{}, xrh_tag);
        else channel_loop(role_tag, std::false_type{}, xrh_tag);
    };
    auto by_half = [&](auto role_tag) {
        if constexpr (NW == 8) {
            by_order(role_tag, std::integral_constant<int, 0>{});
        } else {
            if (wave & 1) by_order(role_tag, std::integral_constant<int, 1>{});
            else by_order(role_tag, std::integral_constant<int, 0>{});
        }
    };
    if constexpr (UPS) {
        if (role == 0) by_half(std::integral_constant<int, 0>{});
        else if (role == 1) by_half(std::integral_constant<int, 1>{});
        else by_half(std::integral_constant<int, 2>{});
    } else {
        by_half(std::integral_constant<int, 0>{});
    }

    WINO_STAMP(4);               // channel loop done
    // ---- output transform: four rounds of 16 output channels through LDS.  Bias, time-embedding and
    // residual operands of all four rounds are fetched up front so their latency is paid once, under
    // the first round's LDS traffic, rather than once per round behind a barrier.
    const size_t HWout = (size_t)p.Hc * p.Wc;
    constexpr int EK = 1024 / G::THREADS;
    const int et = tid & 63;                                   // tile of this thread (same in every round)
    const int eimg = et / (TY * TX), ety = (et / TX) % TY, etx = et % TX;
    const int eb = b0 + eimg, eoy = oy0 + 2 * ety, eox = ox0 + 2 * etx;
    const int ebc = min(eb, p.B - 1);
    int eoff[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) eoff[i][j] = min(eoy + i, p.Hc - 1) * p.Wc + min(eox + j, p.Wc - 1);
    // K-split: every share stores its raw partial sums; bias, residual, ReLU and statistics happen in the reduction
    float* const edst = p.part ? p.part + (size_t)ks * p.B * p.Cout * HWout : p.out;
    // One image per workgroup: the output channel of a thread is wave-uniform in every round, so plane base pointers,
    // bias and embedding are scalar and the per-lane part of every address is the same four 32-bit pixel offsets.
    // (Several images per workgroup: the image, and with it the plane, varies along the lanes.)
    constexpr bool SCALAR_PLANE = NIMG == 1;
    auto round_co = [&](int q, int k) {       // output channel handled in round q, sub-round k (scalar if SCALAR_PLANE)
        const int w16 = SCALAR_PLANE ? wave_u + k * NW : ((tid + k * G::THREADS) >> 6);
        return co0 + (q >> 1) * 32 + 16 * (q & 1) + w16;
    };
    // Rows of even length: a thread's two pixels of a row (ox even) are inside or outside together and 8-byte aligned
    // (planes are 16-byte aligned multiples of the row length), so they move as one float2.  For the clamped
    // out-of-image tiles eoff[i][0] is then the last pixel of the row: the pair would straddle the row end, so those
    // lanes read/write nothing through the pair path (pin is false; the residual value read is never used).
    const bool pair_ok = (p.Wc & 1) == 0 && (eox + 1 < p.Wc);
    float eadd[4][EK], eres[4][EK][2][2];
#pragma unroll
    for (int q = 0; q < ((WINO_ABLATE & 2) ? 0 : 4); ++q) {
#pragma unroll
        for (int k = 0; k < EK; ++k) {
            const int coc = min(round_co(q, k), p.Cout - 1);
            float add = 0.0f;
            if (p.bias && !p.part) add += p.bias[coc];
            if (p.chan_bias && !p.part) add += p.chan_bias[(size_t)ebc * p.chan_bias_stride + coc];
            eadd[q][k] = add;
            const size_t plane = ((size_t)ebc * p.Cout + coc) * HWout;
            gfptr_t rb = (gfptr_t)(p.residual + plane);
            if constexpr (SCALAR_PLANE) rb = (gfptr_t)uniform_ptr(p.residual + plane);
            if (p.residual && !p.part) {
                if (pair_ok) {          // even row length: the thread's two pixels of a row are one aligned float2
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const v2f_t t = *(gf2ptr_t)(rb + eoff[i][0]);
                        eres[q][k][i][0] = t.x;
                        eres[q][k][i][1] = t.y;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) eres[q][k][i][j] = rb[eoff[i][j]];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) eres[q][k][i][j] = 0.0f;
            }
        }
    }
    // GroupNorm partials: pixels of this workgroup tile inside the image (the same for every channel and image)
    const float tile_cnt = (float)(min(2 * TY, p.Hc - oy0) * min(2 * TX, p.Wc - ox0));
    bool pin[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) pin[i][j] = eoy + i < p.Hc && eox + j < p.Wc;
    WINO_STAMP(5);               // epilogue operands requested
#pragma unroll
    for (int q = 0; q < ((WINO_ABLATE & 2) ? 0 : 4); ++q) {
        const int mt = q >> 1, rbase = 8 * (q & 1);
        if (role == 0) {
#pragma unroll
            for (int x = 0; x < G::XPW; ++x) {
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) {
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        const int r = rbase + rr;
                        const int row16 = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;     // row within the 16-channel block
                        M_lds[((xi_w + x) * 16 + row16) * 64 + nn * 32 + l31] = acc[x][mt][nn][r];
                    }
                }
            }
        } else if (UPS && role == 1 && mt == qm) {             // this wave's tile of the shared position
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = rbase + rr;
                const int row16 = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;
                M_lds[(15 * 16 + row16) * 64 + qn * 32 + l31] = acc[0][0][0][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < EK; ++k) {
            const int co16 = (tid + k * G::THREADS) >> 6;       // 1024 (channel, tile) pairs per round (LDS row)
            float m[4][4];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi)
                m[xi >> 2][xi & 3] = (UPS && ((WINO_UPS_ZERO >> xi) & 1u)) ? 0.0f : M_lds[(xi * 16 + co16) * 64 + et];
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
            const int co = round_co(q, k);
            const bool ok = co < p.Cout && eb < p.B;
            const size_t plane = ((size_t)ebc * p.Cout + min(co, p.Cout - 1)) * HWout;
            gfwptr_t db = (gfwptr_t)(edst + plane);
            if constexpr (SCALAR_PLANE) db = (gfwptr_t)uniform_ptr(edst + plane);
            float vv[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v = y[i][j] + eadd[q][k] + eres[q][k][i][j];
                    if (p.relu && !p.part) v = fmaxf(v, 0.0f);
                    if (!pair_ok && ok && pin[i][j]) db[eoff[i][j]] = v;
                    vv[i][j] = v;
                }
                if (pair_ok && ok && pin[i][0]) *(gf2wptr_t)(db + eoff[i][0]) = v2f_t{vv[i][0], vv[i][1]};
            }
            if (!WINO_TIMING && p.stats && !p.part) {
                // One output channel per wave here; lanes are the 64 tiles: all of one image (NIMG == 1) or 16 per
                // image.  Per image and workgroup: (count, sum, M2).  Single pass with the sums shifted by one of
                // the tile's own values K (its first pixel, always inside the image): M2 = sum d^2 - (sum d)^2 / n with
                // d = v - K, exact algebra and free of cancellation because |mean - K| is of the order of the
                // spread; gn_finalize_kernel merges the partials.
                const int slots = p.groups_x * p.groups_y, slot = gy * p.groups_x + gx;
                float K;
                if constexpr (NIMG == 1) K = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv[0][0]), 0));
                else K = __shfl(vv[0][0], lane & ~(TY * TX - 1));
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float d = pin[i][j] ? vv[i][j] - K : 0.0f;
                        s1 += d;
                        s2 = fmaf(d, d, s2);
                    }
                s1 = (NIMG == 1) ? wave64_sum(s1) : row16_sum(s1);
                s2 = (NIMG == 1) ? wave64_sum(s2) : row16_sum(s2);
                if (ok && (et % (TY * TX)) == 0) {
                    float4* dst = reinterpret_cast<float4*>(p.stats) + ((size_t)eb * p.Cout + co) * slots + slot;
                    *dst = make_float4(tile_cnt, fmaf(tile_cnt, K, s1), fmaxf(s2 - s1 * s1 / tile_cnt, 0.0f), 0.0f);
                }
            }
        }
        __syncthreads();
    }
    WINO_STAMP(6);               // output transform issued
#if WINO_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WINO_STAMP(7);               // stores retired
#endif
}

// U = G g G^T per (co, ci), float64 arithmetic, written as [Cin_pad][16][cout_pad] (zero padded)
__global__ void winograd_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int cin_pad, int cout_pad,
                                     float* __restrict__ out) {
    const size_t total = (size_t)cin_pad * cout_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        winograd_pack_elem(i, w, Cout, Cin, cin_pad, cout_pad, out);
    }
}

// two packings back to back: [Cin_pad][16][cout_pad] (first form) and [chunk][xi][32-ch block][lane][4] (wide form, its
// channel blocks padded to whole 128-channel tiles)
static int64_t winograd_first_numel(int Cout, int Cin) { return winograd_first_floats(Cout, Cin); }
static int64_t winograd_wide_numel(int Cout, int Cin) { return (int64_t)round_up(Cin, 16) * 16 * round_up(Cout, 128); }
__global__ void winograd_pack_wide_kernel(const float* __restrict__ u_first, int cin_pad, int cout_pad, int cout_pad128,
                                          float* __restrict__ out);

int launch_winograd_pack_bf3(sisic_ctx*, int Cout, int Cin, float* packed, hipStream_t s);
int launch_winograd_pack(sisic_ctx*, const float* w, int Cout, int Cin, float* packed, hipStream_t s) {
    const int cin_pad = round_up(Cin, 16), cout_pad = conv_cout_pad(Cout);
    const size_t total = (size_t)cin_pad * cout_pad;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(winograd_pack_kernel, dim3(blocks), dim3(256), 0, s, w, Cout, Cin, cin_pad, cout_pad, packed);
    const int cout_pad128 = round_up(Cout, 128);
    const size_t total_w = (size_t)cin_pad * 16 * cout_pad128;
    hipLaunchKernelGGL(winograd_pack_wide_kernel, dim3((unsigned)std::min<size_t>((total_w + 255) / 256, 4096)), dim3(256), 0, s,
                       packed, cin_pad, cout_pad, cout_pad128, packed + winograd_first_numel(Cout, Cin));
    SISIC_HIP(hipGetLastError());
    return launch_winograd_pack_bf3(nullptr, Cout, Cin, packed, s);
}

// third region: the split filters of the bf16x3 form (conv_winograd_bf3.inc): three bf16 terms = 1.5 dwords per filter value
static int64_t winograd_bf3_numel(int Cout, int Cin) { return winograd_wide_numel(Cout, Cin) + winograd_wide_numel(Cout, Cin) / 2; }
int64_t winograd_packed_numel(int Cout, int Cin) {
    return winograd_first_numel(Cout, Cin) + winograd_wide_numel(Cout, Cin) + winograd_bf3_numel(Cout, Cin);
}

#include "conv_winograd_wide.inc"
#include "conv_winograd_col.inc"
#include "conv_winograd_bf3.inc"

int launch_winograd_pack_bf3(sisic_ctx*, int Cout, int Cin, float* packed, hipStream_t s) {
    const int cin_pad = round_up(Cin, 16), cout_pad = conv_cout_pad(Cout), cout_pad128 = round_up(Cout, 128);
    const size_t total = (size_t)winograd_bf3_numel(Cout, Cin);
    unsigned* dst = reinterpret_cast<unsigned*>(packed + winograd_first_numel(Cout, Cin) + winograd_wide_numel(Cout, Cin));
    hipLaunchKernelGGL(winograd_pack_bf3_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, packed, cin_pad,
                       cout_pad, cout_pad128, dst);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// K-split reduction: out = sum_k part[k] + bias + per-sample channel bias + residual (+ReLU), and the GroupNorm partials
// of the result.  One wave per (image, channel) plane of HW <= 256 pixels; the plane is the only statistics slot.
template <int KSPLIT>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ part, int planes, int HW,
                                                            int Cout, const float* __restrict__ bias,
                                                            const float* __restrict__ chan_bias, int chan_bias_stride,
                                                            const float* residual, int relu,     // (out may alias residual: sisic.h)
                                                            float* out, float* __restrict__ stats) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (plane >= planes) return;
    const int b = plane / Cout, co = plane % Cout;
    float add = bias ? bias[co] : 0.0f;
    if (chan_bias) add += chan_bias[(size_t)b * chan_bias_stride + co];
    const size_t base = (size_t)plane * HW, kstride = (size_t)planes * HW;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float s1 = 0.0f, cnt = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (64 * i >= HW) break;               // uniform: planes of 64 pixels take one round
        const int px = lane + 64 * i;
        const bool in = px < HW;
        const size_t o = base + min(px, HW - 1);
        float pk[KSPLIT];
#pragma unroll
        for (int k = 0; k < KSPLIT; ++k) pk[k] = part[o + k * kstride];      // all in flight together
        const float r = residual ? residual[o] : 0.0f;
        float acc = pk[0];
#pragma unroll
        for (int k = 1; k < KSPLIT; ++k) acc += pk[k];
        acc += add;
        acc += r;
        if (relu) acc = fmaxf(acc, 0.0f);
        if (in) {
            out[o] = acc;
            s1 += acc;
            cnt += 1.0f;
        }
        v[i] = acc;
    }
    if (stats) {
        s1 = wave64_sum(s1);
        cnt = wave64_sum(cnt);
        const float mean = s1 / fmaxf(cnt, 1.0f);
        float m2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = v[i] - mean;
            if (lane + 64 * i < HW) m2 += d * d;
        }
        m2 = wave64_sum(m2);
        if (lane == 0) reinterpret_cast<float4*>(stats)[plane] = make_float4(cnt, s1, m2, 0.0f);
    }
}

// The same reduction for planes of exactly 64 pixels (the 8x8 level: every launch of the headline model) with 16 lanes per plane
// and 16-byte accesses: a quarter of the waves and of the memory instructions.  Output values: the same sums in the same order,
// bit for bit; the GroupNorm partials are summed over another tree (four values per lane, then a 16-lane butterfly).
// FIN (round 4): the launch also finalizes the GroupNorm over its own output when a group is EIGHT channels (the 8x8 level's 256
// channels in 32 groups): a workgroup's 16 planes are two whole groups, so the partials of a group meet in LDS and every
// channel's thread merges them -- in the order and with the operations of gn_finalize_kernel (gn_merge.h: the same bits) --
// and writes its (scale, shift).  One 5 us launch less per such GroupNorm.
__device__ __forceinline__ bool valid_wg(unsigned wg, int planes) { return (int)(wg * 16u) < planes; }
struct ReduceFin {
    const float* gamma;
    const float* beta;
    float* scale;
    float* shift;
    float* mean_rstd;
    int groups;
    float eps;
};
template <bool FIN>
__global__ void __launch_bounds__(256) splitk_reduce64_kernel(const float4* __restrict__ part, int planes, int Cout,
                                                              const float* __restrict__ bias, const float* __restrict__ chan_bias,
                                                              int chan_bias_stride, const float4* residual, int relu,   // (out may alias residual)
                                                              float4* out, float4* __restrict__ stats, const ReduceFin fin) {
    const int plane_raw = blockIdx.x * 16 + (threadIdx.x >> 4), l16 = threadIdx.x & 15;
    const bool valid = plane_raw < planes;               // (a row without a plane runs along on plane 0: the butterflies need every lane)
    const int plane = valid ? plane_raw : 0;
    const int b = plane / Cout, co = plane % Cout;
    float add = bias ? bias[co] : 0.0f;
    if (chan_bias) add += chan_bias[(size_t)b * chan_bias_stride + co];
    const size_t o = (size_t)plane * 16 + l16, kstride = (size_t)planes * 16;
    float4 pk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pk[k] = part[o + k * kstride];               // all in flight together
    const float4 r = residual ? residual[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float a[4] = {pk[0].x, pk[0].y, pk[0].z, pk[0].w};
    const float rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] += (e == 0 ? pk[1].x : e == 1 ? pk[1].y : e == 2 ? pk[1].z : pk[1].w);
        a[e] += (e == 0 ? pk[2].x : e == 1 ? pk[2].y : e == 2 ? pk[2].z : pk[2].w);
        a[e] += (e == 0 ? pk[3].x : e == 1 ? pk[3].y : e == 2 ? pk[3].z : pk[3].w);
        a[e] += add;
        a[e] += rr[e];
        if (relu) a[e] = fmaxf(a[e], 0.0f);
    }
    if (valid) out[o] = make_float4(a[0], a[1], a[2], a[3]);
    if (stats) {
        const float s1 = row16_sum((a[0] + a[1]) + (a[2] + a[3]));
        const float mean = s1 * (1.0f / 64.0f);
        float m2 = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = a[e] - mean; m2 += d * d; }
        m2 = row16_sum(m2);
        if (valid && l16 == 0) stats[plane] = make_float4(64.0f, s1, m2, 0.0f);
    }
    if constexpr (FIN) {
        // (launcher: Cout a multiple of 16 -- a workgroup's planes are one image's channels co0 .. co0 + 15 -- and 8 channels per group)
        __shared__ float2 part_sm[16];
        const float s1 = row16_sum((a[0] + a[1]) + (a[2] + a[3]));          // (as above: the partial this plane would leave)
        const float mean_p = s1 * (1.0f / 64.0f);
        float m2 = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = a[e] - mean_p; m2 += d * d; }
        m2 = row16_sum(m2);
        const int row = threadIdx.x >> 4;
        if (l16 == 0) part_sm[row] = make_float2(s1, m2);
        __syncthreads();
        if (threadIdx.x < 16 && valid_wg(blockIdx.x, planes)) {
            const int ch = threadIdx.x, g8 = ch >> 3;
            // gn_finalize_kernel's lanes 0 .. 7 hold the group's eight partials and sum them through its 64-lane butterfly:
            // ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)), in float64
            double pn[8], ps[8], pm[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { pn[i] = (double)64.0f + (double)0.0f; ps[i] = (double)part_sm[8 * g8 + i].x + (double)0.0f; pm[i] = (double)part_sm[8 * g8 + i].y + (double)0.0f; }
            auto tree = [](const double (&v)[8]) { return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])); };
            const double n = tree(pn), sum1 = tree(ps), sum2 = tree(pm);
            const double mean = sum1 / n;
            double bt[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) bt[i] = 0.0 + gn_between_term(64.0f, part_sm[8 * g8 + i].x, mean);
            const double between = tree(bt);
            float meanf, rstd;
            gn_mean_rstd(n, sum1, sum2, between, fin.eps, meanf, rstd, mean);
            const int plane0 = blockIdx.x * 16, b = plane0 / Cout, co = plane0 % Cout + ch;
            float sc, sh;
            gn_affine(fin.gamma[co], fin.beta[co], meanf, rstd, sc, sh);
            fin.scale[(size_t)b * Cout + co] = sc;
            fin.shift[(size_t)b * Cout + co] = sh;
            if (fin.mean_rstd && (ch & 7) == 0) {
                const int g = co >> 3;
                fin.mean_rstd[2 * ((size_t)b * fin.groups + g)] = meanf;
                fin.mean_rstd[2 * ((size_t)b * fin.groups + g) + 1] = rstd;
            }
        }
    }
}

// K-split reduction for planes of any size (latency mode of the second geometry): one workgroup per 1024-pixel segment of
// an (image, channel) plane; every segment is one statistics slot.  Fixed summation order over the K partial slabs and
// inside the block.
__global__ void __launch_bounds__(256) splitk_reduce_plane_kernel(const float* __restrict__ part, int ksplit, int planes, int HW,
                                                                  int segs, int Cout, const float* __restrict__ bias,
                                                                  const float* __restrict__ chan_bias, int chan_bias_stride,
                                                                  const float* residual, int relu,      // (out may alias residual)
                                                                  float* out, float* __restrict__ stats) {
    __shared__ float red[8];
    const int plane = blockIdx.x / segs, seg = blockIdx.x % segs;
    const int b = plane / Cout, co = plane % Cout;
    float add = bias ? bias[co] : 0.0f;
    if (chan_bias) add += chan_bias[(size_t)b * chan_bias_stride + co];
    const size_t base = (size_t)plane * HW, kstride = (size_t)planes * HW;
    const int p0 = seg * 1024, p1 = min(HW, p0 + 1024);
    auto value = [&](int px) {
        float acc = part[base + px];
        for (int k = 1; k < ksplit; ++k) acc += part[base + px + k * kstride];
        acc += add;
        if (residual) acc += residual[base + px];
        return relu ? fmaxf(acc, 0.0f) : acc;
    };
    const float K0 = value(p0);                    // shift of the running sums: one of the segment's own values
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int px = p0 + threadIdx.x + 256 * i;
        if (px < p1) {
            const float v = value(px);
            out[base + px] = v;
            const float d = v - K0;
            s1 += d;
            s2 = fmaf(d, d, s2);
        }
    }
    if (stats) {
        s1 = wave64_sum(s1);
        s2 = wave64_sum(s2);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) { red[wave] = s1; red[4 + wave] = s2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t1 = (red[0] + red[1]) + (red[2] + red[3]), t2 = (red[4] + red[5]) + (red[6] + red[7]);
            const float cnt = (float)(p1 - p0);
            reinterpret_cast<float4*>(stats)[(size_t)plane * segs + seg] =
                make_float4(cnt, fmaf(cnt, K0, t1), fmaxf(t2 - t1 * t1 / cnt, 0.0f), 0.0f);
        }
    }
}

template <int NIMG, int TY, int TX, int PRO, int NW, bool UPS = false>
static int launch_wino(sisic_ctx* ctx, WinoParams& p, hipStream_t s) {
    using G = WinoGeom<NIMG, TY, TX, NW, UPS>;
    p.groups_x = cdiv(p.Wc, 2 * TX);
    p.groups_y = cdiv(p.Hc, 2 * TY);
    p.groups_b = cdiv(p.B, NIMG);
    p.n_co_tiles = p.cout_pad / W_CO;
    p.nchunks = cdiv(p.c0 + p.c1, W_CIC);
    if (p.ksplit < 1) p.ksplit = 1;
    SISIC_REQUIRE(p.nchunks % p.ksplit == 0, "conv2d(winograd): %d channel chunks do not split %d ways", p.nchunks, p.ksplit);
    const int64_t nwg = (int64_t)p.groups_x * p.groups_y * p.groups_b * p.n_co_tiles * p.ksplit;
    SISIC_REQUIRE(nwg > 0 && nwg < (int64_t(1) << 31), "conv2d(winograd): grid too large");
    p.nwg = (int)nwg;
    auto kern = conv_winograd_kernel<NIMG, TY, TX, PRO, NW, UPS>;
    static std::atomic<uint64_t> lds_opt_in{0};     // one bit per device (common.h)
    SISIC_TRY(ensure_dynamic_lds(ctx, reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_opt_in));
    hipLaunchKernelGGL(kern, dim3(p.nwg), dim3(G::THREADS), G::LDS_BYTES, s, p);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

template <int NIMG, int TY, int TX, int NW>
static int launch_wino_pro(sisic_ctx* ctx, WinoParams& p, hipStream_t s) {
    const int pro = (p.gn_scale == nullptr) ? 0 : (p.gn_silu ? 2 : 1);
    if constexpr (NIMG == 1 && NW == 16) {
        // nearest-2x input: the nine-position form (cfg 62 keeps the generic kernel for comparison)
        if (p.ups && p.stagger && pro == 0) return launch_wino<NIMG, TY, TX, 0, NW, true>(ctx, p, s);
    }
    if (pro == 2) return launch_wino<NIMG, TY, TX, 2, NW>(ctx, p, s);
    if (pro == 1) return launch_wino<NIMG, TY, TX, 1, NW>(ctx, p, s);
    return launch_wino<NIMG, TY, TX, 0, NW>(ctx, p, s);
}

// tile_cfg 60: 1 image x 8x8 tiles (16x16 output pixels), 8 waves;  61: 4 images x 4x4 tiles (8x8 outputs each), 8 waves;
//          62 / 63: the same two tilings with 16 waves (one transform position per wave, 4 waves per SIMD)
//          64..67 = 60..63 with the MFMA-first / stage-first phase stagger between SIMD partner waves
//          68 / 69: second geometry, 32 tiles per workgroup, filters straight from global memory into registers:
//              68 = 128 output channels x 16 waves (Cout > 64), 69 = 64 channels x 8 waves, two workgroups per CU
//          78 / 79: 68 / 69 in latency mode: input channels K-split so that one image fills the chip (wino_latency_ksplit)
//          (68 / 69 / 78 / 79 / 91 run the THIRD form, conv_winograd_col.inc -- same tiles, same bits, a wave owns a column of
//           the position grid -- unless SISIC_WINO_COL=0;  70 / 71 force the third form, 72 / 73 the second: A/B and tests)
//          90: tiling 67 with the input channels split over four workgroups per tile + splitk_reduce_kernel -- for the
//              8x8 level, where 64 tiles x 64 channels per workgroup leave 3/4 of the CUs without work
//          91: the same split on the second geometry: 128 channels x (two images x 16 tiles) per workgroup; bit-identical to 90
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
    p.stats = a.stats_out;
    SISIC_REQUIRE(4.0 * std::max(a.c0, a.c1) * a.Hin * a.Win * ((cfg == 61 || cfg == 63 || cfg == 65 || cfg == 67 || cfg == 90) ? 4 : 1) < 4294967296.0,
                  "conv2d(winograd): per-thread load offsets are 32-bit; this tensor needs the direct kernel");
    p.stagger = ((cfg >= 64 && cfg <= 67) || cfg == 90) ? 1 : 0;
    p.ksplit = 1;
    static const bool col_default = [] { const char* e = std::getenv("SISIC_WINO_COL"); return !e || std::atoi(e) != 0; }();
    if (cfg == 74 && conv_finalizes(a)) {       // (one 16x16-pixel tile per image: conv_mfma.hip)
        SISIC_REQUIRE(a.fin_beta && a.fin_scale && a.fin_shift, "conv2d: fin_gamma given without fin_beta / fin_scale / fin_shift");
        p.fin_gamma = a.fin_gamma; p.fin_beta = a.fin_beta; p.fin_scale = a.fin_scale; p.fin_shift = a.fin_shift;
        p.fin_mean_rstd = a.fin_mean_rstd; p.fin_groups = a.fin_groups; p.fin_eps = a.fin_eps;
    }
    if (cfg == 74) {                // fp32-equivalent products on the bf16 pipe (conv_winograd_bf3.inc)
        // (a nearest-2x input is read through the staging plan's addresses: all 16 positions are multiplied, where the f32
        //  upsample form multiplies 9 -- which of the two is faster depends on the plane, conv_mfma.hip)
        p.uw = u_packed + winograd_first_numel(a.Cout, a.c0 + a.c1) + winograd_wide_numel(a.Cout, a.c0 + a.c1);
        p.cout_pad = round_up(a.Cout, 128);
        return launch_bf3_pro(ctx, p, s);
    }
    if ((cfg >= 68 && cfg <= 73) || cfg == 78 || cfg == 79) {           // second / third geometry (conv_winograd_wide.inc, _col.inc)
        SISIC_REQUIRE(!p.ups, "conv2d(winograd wide): no upsample form");
        p.uw = u_packed + winograd_first_numel(a.Cout, a.c0 + a.c1);
        p.cout_pad = round_up(a.Cout, 128);
        const bool wide = cfg == 68 || cfg == 78 || cfg == 70 || cfg == 72;
        const bool col = (cfg == 70 || cfg == 71) ? true : ((cfg == 72 || cfg == 73) ? false : col_default);
        const int K = (cfg == 78 || cfg == 79) ? wino_latency_ksplit(a.Cout, a.c0 + a.c1, p.Hc, p.Wc) : 1;
        auto launch = [&]() -> int {
            if (col) return wide ? launch_col_pro<128, 16>(ctx, p, s) : launch_col_pro<64, 8>(ctx, p, s);
            return wide ? launch_wide_pro<128, 16>(ctx, p, s) : launch_wide_pro<64, 8>(ctx, p, s);
        };
        if (K == 1) return launch();
        // latency mode: the input channels split K ways over workgroups, partial slabs summed by the plane reduction
        const size_t HW = (size_t)p.Hc * p.Wc, planes = (size_t)a.B * a.Cout;
        const size_t need = (size_t)K * planes * HW;
        float* scratch = nullptr;
        {
            std::lock_guard<std::mutex> lock(ctx->splitk_mutex);
            auto& buf = ctx->splitk[s];
            if (buf.floats < need) {
                SISIC_HIP(hipStreamSynchronize(s));
                if (buf.p) SISIC_HIP(hipFree(buf.p));
                buf.p = nullptr; buf.floats = 0;
                SISIC_HIP(hipMalloc(reinterpret_cast<void**>(&buf.p), need * sizeof(float)));
                buf.floats = need;
                ctx->scratch_generation.fetch_add(1);
            }
            scratch = buf.p;
        }
        p.ksplit = K;
        p.part = scratch;
        SISIC_TRY(launch());
        const int segs = wino_latency_segments(p.Hc, p.Wc);
        hipLaunchKernelGGL(splitk_reduce_plane_kernel, dim3((unsigned)(planes * segs)), dim3(256), 0, s, scratch, K, (int)planes,
                           (int)HW, segs, a.Cout, a.bias, a.chan_bias, a.chan_bias_stride, a.residual, a.relu, a.out, a.stats_out);
        SISIC_HIP(hipGetLastError());
        return SISIC_OK;
    }
    if (cfg == 90 || cfg == 91 || cfg == 92) {
        constexpr int K = 4;
        const size_t HW = (size_t)p.Hc * p.Wc, planes = (size_t)a.B * a.Cout;
        SISIC_REQUIRE(HW <= 256 && (cdiv(a.c0 + a.c1, W_CIC) % K) == 0,
                      "conv2d(winograd K-split): needs <= 256 output pixels per image and a multiple of %d input channels", K * W_CIC);
        const size_t need = (size_t)K * planes * HW;
        float* scratch = nullptr;
        {
            std::lock_guard<std::mutex> lock(ctx->splitk_mutex);
            auto& buf = ctx->splitk[s];
            if (buf.floats < need) {
                SISIC_HIP(hipStreamSynchronize(s));          // earlier launches on this stream may still read the old buffer
                if (buf.p) SISIC_HIP(hipFree(buf.p));
                buf.p = nullptr; buf.floats = 0;
                SISIC_HIP(hipMalloc(reinterpret_cast<void**>(&buf.p), need * sizeof(float)));
                buf.floats = need;
                ctx->scratch_generation.fetch_add(1);
            }
            scratch = buf.p;
        }
        p.ksplit = K;
        p.part = scratch;
        if (cfg == 92) {          // bf16x3 products, four images per workgroup (conv_winograd_bf3.inc)
            SISIC_REQUIRE(!p.ups && p.Hc <= 8 && p.Wc <= 8, "conv2d(winograd bf16x3, 8x8): plain stride-1 convolutions of at most 8x8 pixels");
            p.uw = u_packed + winograd_first_numel(a.Cout, a.c0 + a.c1) + winograd_wide_numel(a.Cout, a.c0 + a.c1);
            p.cout_pad = round_up(a.Cout, 128);
            SISIC_TRY(launch_bf3q_pro(ctx, p, s));
        } else if (cfg == 91) {   // second geometry, two images per workgroup (conv_winograd_wide.inc, PAIR)
            SISIC_REQUIRE(!p.ups && p.Hc <= 8 && p.Wc <= 8, "conv2d(winograd wide, image pairs): plain stride-1 convolutions of at most 8x8 pixels");
            p.uw = u_packed + winograd_first_numel(a.Cout, a.c0 + a.c1);
            p.cout_pad = round_up(a.Cout, 128);
            if (col_default) SISIC_TRY((launch_col_pro<128, 16, true>(ctx, p, s)));
            else SISIC_TRY((launch_wide_pro<128, 16, true>(ctx, p, s)));
        } else {
            SISIC_TRY((launch_wino_pro<4, 4, 4, 16>(ctx, p, s)));
        }
        const bool al16 = ((reinterpret_cast<uintptr_t>(scratch) | reinterpret_cast<uintptr_t>(a.residual) | reinterpret_cast<uintptr_t>(a.out) |
                            reinterpret_cast<uintptr_t>(a.stats_out)) & 15) == 0;
        if (HW == 64 && al16) {
            const bool fin = conv_finalizes(a);
            const ReduceFin rf{a.fin_gamma, a.fin_beta, a.fin_scale, a.fin_shift, a.fin_mean_rstd, a.fin_groups, a.fin_eps};
            if (fin) {
                SISIC_REQUIRE(a.fin_beta && a.fin_scale && a.fin_shift, "conv2d: fin_gamma given without fin_beta / fin_scale / fin_shift");
                hipLaunchKernelGGL(splitk_reduce64_kernel<true>, dim3((unsigned)((planes + 15) / 16)), dim3(256), 0, s,
                                   reinterpret_cast<const float4*>(scratch), (int)planes, a.Cout, a.bias, a.chan_bias, a.chan_bias_stride,
                                   reinterpret_cast<const float4*>(a.residual), a.relu, reinterpret_cast<float4*>(a.out),
                                   reinterpret_cast<float4*>(a.stats_out), rf);
            } else {
                hipLaunchKernelGGL(splitk_reduce64_kernel<false>, dim3((unsigned)((planes + 15) / 16)), dim3(256), 0, s,
                                   reinterpret_cast<const float4*>(scratch), (int)planes, a.Cout, a.bias, a.chan_bias, a.chan_bias_stride,
                                   reinterpret_cast<const float4*>(a.residual), a.relu, reinterpret_cast<float4*>(a.out),
                                   reinterpret_cast<float4*>(a.stats_out), rf);
            }
        }
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, s, scratch,
                               (int)planes, (int)HW, a.Cout, a.bias, a.chan_bias, a.chan_bias_stride, a.residual, a.relu,
                               a.out, a.stats_out);
        SISIC_HIP(hipGetLastError());
        return SISIC_OK;
    }
    switch (cfg) {
        case 64: return launch_wino_pro<1, 8, 8, 8>(ctx, p, s);
        case 65: return launch_wino_pro<4, 4, 4, 8>(ctx, p, s);
        case 66: return launch_wino_pro<1, 8, 8, 16>(ctx, p, s);
        case 67: return launch_wino_pro<4, 4, 4, 16>(ctx, p, s);
        case 61: return launch_wino_pro<4, 4, 4, 8>(ctx, p, s);
        case 62: return launch_wino_pro<1, 8, 8, 16>(ctx, p, s);
        case 63: return launch_wino_pro<4, 4, 4, 16>(ctx, p, s);
        default: return launch_wino_pro<1, 8, 8, 8>(ctx, p, s);
    }
}

}  // namespace sisic
