// train_kernels.hip -- the kernels of the UNet training step that the sampling path does not have.
//
// Reference: diffusion/train_diffusion.py:201-266 -- per batch
//     noise = randn_like(images); t = randint(0, 1000, (B,)); noisy = scheduler.add_noise(images, noise, t)       :215-217
//     loss = mse_loss(model(noisy, t).sample, noise)                                                                :218-219
//     scaler.scale(loss).backward(); scaler.step(Adam(lr=1e-4)); scaler.update()                                   :230-240
// torch.autograd does the backward pass there.  Here it is explicit:
//   * backward-DATA of every convolution is a sisic_conv2d launch with the transposed, tap-flipped filter (the forward
//     kernels: Winograd F(2x2,3x3) / direct MFMA; stride 2 = zero-insertion form) -- nothing new in this file;
//   * backward-WEIGHT is conv_wgrad_kernel below: an implicit GEMM  dW[co, (ci,tap)] = sum_pixels dy[co,p] * a[ci, p+tap]
//     on v_mfma_f32_32x32x2_f32 with the PIXELS as the contraction dimension, the forward's GroupNorm+SiLU prologue
//     recomputed while the input halo is staged (the normalised activation is never stored), K-split over pixel blocks
//     with a fixed-order reduction (no atomics: bit-reproducible gradients);
//   * GroupNorm(+SiLU) backward, attention backward (recomputing softmax per head, d = 8), the small linears of the time
//     embedding, MSE, add_noise and Adam are HBM- or latency-bound vector kernels.
#include "common.h"
#include "pack_device.h"
#include "train.h"

namespace sisic {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum_t(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for 256-thread blocks, fixed order
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum_t(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float sigmoid_acc(float v) { return 1.0f / (1.0f + expf(-v)); }
// d silu(u) / du = s (1 + u (1 - s)),  s = sigmoid(u)
__device__ __forceinline__ float silu_grad(float u) {
    const float s = sigmoid_acc(u);
    return s * (1.0f + u * (1.0f - s));
}
__device__ __forceinline__ float silu_fwd(float u) { return u * __builtin_amdgcn_rcpf(1.0f + __expf(-u)); }   // as the forward kernels

// ================================================================ convolution backward-weight ======================
// Workgroup = 64 output channels x 64 input channels x KK taps of dW, summed over this workgroup's share of the pixel
// blocks (TR x TC output pixels each).  Four waves: wave (mt, nt) owns the 32x32 block (co half mt, ci half nt) for all
// taps -- KK accumulators of 16 registers, two workgroups per CU.  (Splitting the nine taps over two wave groups, 80
// accumulator registers and four waves per SIMD, was tried: the prefetch does not fit 128 registers beside them.)  Per pixel block: dy tile [64][P] and the input halo [64][IH*IW]
// (prologue applied, zero padding after it, concat / nearest-2x index maps as in the forward) are staged in LDS; a k-step
// of the MFMA is two neighbouring pixels; the A fragment (dy) is shared by a wave's taps, the B fragment is the halo read
// at the tap's offset.
//   * the halo loads of block i+1 are issued before the MFMAs of block i and consumed after them (register prefetch);
//   * the partial tile leaves through LDS, one 32x32xKK quadrant at a time, so that a wave stores runs of 288 contiguous
//     floats instead of 144 scattered dwords per lane.
// Measured against round 2's first version (load, stage, compute one block after the other; scattered stores), batch 32 at
// 64x64: 1x1 74 -> 43 us per launch; 3x3 unchanged within 3 % -- those launches are bounded by their K-split slabs (512
// workgroups x a 147-KB tile = 75 MB written and re-read per weight gradient of a 64-channel layer), not by load latency.
// Every (co, ci, tap) element still sums its pixels in the same order: the gradients keep their bits.
struct WgradParams {
    const float* in0; const float* in1; int c0, c1;
    int B, Hin, Win, ups, Hc, Wc, Hout, Wout;
    const float* gn_scale; const float* gn_shift; int gn_silu;
    const float* dy; int Cout;
    float* part;               // [ksplit][Cout][Cin][KK]
    int ksplit, n_co_tiles, n_ci_tiles, tiles_x, tiles_y, nblocks, blocks_per_slice;
};

template <int KS, int STRIDE, int TR, int TC>
struct WgradGeom {
    static constexpr int KK = KS * KS;
    static constexpr int PAD = KS / 2;
    static constexpr int P = TR * TC;                       // output pixels per block
    static constexpr int IH = (TR - 1) * STRIDE + KS, IW = (TC - 1) * STRIDE + KS;
    static constexpr int HEL = IH * IW;
    static constexpr int CHS = HEL | 1;                     // odd channel stride: the 32 lanes of a fragment read hit 32 banks
    static constexpr int DYS = P + 1;
    static constexpr int TG = 1;                            // tap groups = wave groups (2 was tried for 3x3: 8 waves x 80 accumulators do not fit 128 registers beside the prefetch)
    static constexpr int NW = 4 * TG, NTHR = 64 * NW;
    static constexpr int TAPS = (KK + TG - 1) / TG;         // taps per wave: 5 (groups of 5 and 4) | 1
    static constexpr int HCH = 64 / NW;                     // halo channels staged per wave
    static constexpr int HPASS = (HEL + 63) / 64;
    static constexpr int DY_CO_STEP = NTHR / P;             // channels covered per pass of the dy staging
    static constexpr int DY_PASSES = 64 / DY_CO_STEP;
    static constexpr int STAGE_FLOATS = 64 * CHS + 64 * DYS;
    static constexpr int QUAD_FLOATS = 32 * 32 * KK;        // one quadrant of the partial tile, [co][ci][tap]
    static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(STAGE_FLOATS > QUAD_FLOATS ? STAGE_FLOATS : QUAD_FLOATS);
    static_assert(P % 2 == 0 && TC % 2 == 0, "two pixels of a row per MFMA k-step");
    static_assert(NTHR % P == 0 && 64 % DY_CO_STEP == 0, "dy staging plan");
    static_assert(HPASS <= 8, "halo valid mask");
};

template <int KS, int STRIDE, int TR, int TC>
__global__ void __launch_bounds__((WgradGeom<KS, STRIDE, TR, TC>::NTHR), 2) conv_wgrad_kernel(const WgradParams p) {
    using G = WgradGeom<KS, STRIDE, TR, TC>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const aL = smem;                     // [64 ci][CHS]
    float* const dL = smem + 64 * G::CHS;       // [64 co][DYS]

    int work = blockIdx.x;
    const int co_t = work % p.n_co_tiles; work /= p.n_co_tiles;
    const int ci_t = work % p.n_ci_tiles; work /= p.n_ci_tiles;
    const int ks = work;
    const int co0 = co_t * 64, ci0 = ci_t * 64;
    const int Cin = p.c0 + p.c1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int tg = wave >> 2, mt = wave & 1, nt = (wave >> 1) & 1;
    const int HWin = p.Hin * p.Win, HWout = p.Hout * p.Wout;

    f32x16 acc[G::TAPS];
#pragma unroll
    for (int t = 0; t < G::TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int blk_lo = ks * p.blocks_per_slice, blk_hi = min(p.nblocks, blk_lo + p.blocks_per_slice);
    const float* a_base = aL + (nt * 32 + l31) * G::CHS + half * STRIDE;
    const float* d_base = dL + (mt * 32 + l31) * G::DYS + half;

    // ---- staging plans, invariant over the pixel blocks (no division inside the block loop)
    const int dpx = tid % G::P, dco = tid / G::P;          // dy tile: a thread keeps one pixel and walks the output channels
    const int dty = dpx / TC, dtx = dpx % TC;
    int hyy[G::HPASS], hxx[G::HPASS];                      // halo: a wave stages HCH channels, its lanes sweep a channel's HEL elements
#pragma unroll
    for (int q = 0; q < G::HPASS; ++q) {
        const int r = min(lane + 64 * q, G::HEL - 1);
        hyy[q] = r / G::IW;
        hxx[q] = r % G::IW;
    }

    // one pixel block's input halo in flight: raw values, no prologue yet.  (The dy tile is NOT prefetched: its 16 values per
    // thread beside 144 accumulators spill, and a spill reload inside the MFMA loop waits for vmcnt(0) -- i.e. for the very
    // loads the prefetch wanted to overlap.  Its loads are issued at the top of stage() and land under the halo's prologue.)
    struct Pre {
        float hv[G::HCH][G::HPASS];
        unsigned hmask;              // bit q: halo element of pass q lies inside the image
        int b, oy0, ox0;
    };
    auto prefetch = [&](int blk, Pre& r) {
        int t = blk;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const int b = t / p.tiles_y;
        const int oy0 = ty * TR, ox0 = tx * TC;
        r.b = b; r.oy0 = oy0; r.ox0 = ox0;
        const int iy0 = oy0 * STRIDE - G::PAD, ix0 = ox0 * STRIDE - G::PAD;
        int goff[G::HPASS];
        r.hmask = 0;
#pragma unroll
        for (int q = 0; q < G::HPASS; ++q) {
            const int y = iy0 + hyy[q], x = ix0 + hxx[q];
            const bool ok = (lane + 64 * q) < G::HEL && y >= 0 && y < p.Hc && x >= 0 && x < p.Wc;
            r.hmask |= (ok ? 1u : 0u) << q;
            goff[q] = (min(max(y, 0), p.Hc - 1) >> p.ups) * p.Win + (min(max(x, 0), p.Wc - 1) >> p.ups);
        }
#pragma unroll
        for (int j = 0; j < G::HCH; ++j) {
            const int cc = min(ci0 + wave * G::HCH + j, Cin - 1);
            const float* plane = cc < p.c0 ? p.in0 + ((size_t)b * p.c0 + cc) * HWin
                                           : p.in1 + ((size_t)b * p.c1 + (cc - p.c0)) * HWin;
#pragma unroll
            for (int q = 0; q < G::HPASS; ++q) r.hv[j][q] = plane[goff[q]];
        }
    };
    auto stage = [&](const Pre& r) {
        // dy tile: 64 channels x P pixels, zero outside the image / past Cout (requested here, written below)
        const int oy = r.oy0 + dty, ox = r.ox0 + dtx;
        const bool pin = oy < p.Hout && ox < p.Wout;
        const float* dsrc = p.dy + (size_t)r.b * p.Cout * HWout + (size_t)min(oy, p.Hout - 1) * p.Wout + min(ox, p.Wout - 1);
        float dyv[G::DY_PASSES];
#pragma unroll
        for (int k = 0; k < G::DY_PASSES; ++k) dyv[k] = dsrc[(size_t)min(co0 + dco + k * G::DY_CO_STEP, p.Cout - 1) * HWout];
        // input halo: 64 channels x IH x IW with the forward's prologue, zero padding AFTER it
#pragma unroll
        for (int j = 0; j < G::HCH; ++j) {
            const int ci = wave * G::HCH + j;
            const int c = ci0 + ci;
            const int cc = min(c, Cin - 1);
            float sc = 1.0f, sh = 0.0f;
            if (p.gn_scale) {
                sc = p.gn_scale[(size_t)r.b * Cin + cc];
                sh = p.gn_shift[(size_t)r.b * Cin + cc];
            }
#pragma unroll
            for (int q = 0; q < G::HPASS; ++q) {
                float v = r.hv[j][q];
                if (p.gn_scale) {
                    v = v * sc + sh;
                    if (p.gn_silu) v = silu_fwd(v);
                }
                if ((lane + 64 * q) < G::HEL) aL[ci * G::CHS + lane + 64 * q] = (((r.hmask >> q) & 1u) && c < Cin) ? v : 0.0f;
            }
        }
#pragma unroll
        for (int k = 0; k < G::DY_PASSES; ++k) {
            const int co = dco + k * G::DY_CO_STEP;
            dL[co * G::DYS + dpx] = (pin && (co0 + co) < p.Cout) ? dyv[k] : 0.0f;
        }
    };
    // P/2 k-steps x this wave's taps [T0, T0 + NT): compile-time tap offsets
    auto compute = [&](auto t0_tag, auto nt_tag) {
        constexpr int T0 = decltype(t0_tag)::value, NTAP = decltype(nt_tag)::value;
#pragma unroll 2
        for (int kp = 0; kp < G::P / 2; ++kp) {
            const int py = (2 * kp) / TC, pxx = (2 * kp) % TC;        // this lane's pixel is (py, pxx + half)
            const float a = d_base[2 * kp];
#pragma unroll
            for (int tt = 0; tt < NTAP; ++tt) {
                constexpr int dummy = 0; (void)dummy;
                const int t = T0 + tt;
                const float bv = a_base[(py * STRIDE + t / KS) * G::IW + pxx * STRIDE + t % KS];
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[tt], 0, 0, 0);
            }
        }
    };

    Pre pre;
    if (blk_lo < blk_hi) prefetch(blk_lo, pre);
    for (int blk = blk_lo; blk < blk_hi; ++blk) {
        __syncthreads();                          // the previous block's fragments have been read
        stage(pre);
        __syncthreads();
        if (blk + 1 < blk_hi) prefetch(blk + 1, pre);      // in flight during the MFMAs below
        if constexpr (G::TG == 1) {
            compute(std::integral_constant<int, 0>{}, std::integral_constant<int, G::KK>{});
        } else {
            if (tg == 0) compute(std::integral_constant<int, 0>{}, std::integral_constant<int, G::TAPS>{});
            else compute(std::integral_constant<int, G::TAPS>{}, std::integral_constant<int, G::KK - G::TAPS>{});
        }
    }

    // ---- partial dW of this slice, one (co half, ci half) quadrant at a time through LDS: [32 co][32 ci][KK]
    float* const dst = p.part + (size_t)ks * p.Cout * Cin * G::KK;
    float* const quad = smem;
    const int t0 = tg * G::TAPS, ntap = min(G::TAPS, G::KK - t0);
#pragma unroll 1
    for (int qd = 0; qd < 4; ++qd) {
        const int qm = qd & 1, qn = qd >> 1;
        __syncthreads();                          // staging buffers (first round) / the previous quadrant have been read
        if (mt == qm && nt == qn) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co_l = 8 * (r >> 2) + 4 * half + (r & 3);
#pragma unroll
                for (int tt = 0; tt < G::TAPS; ++tt)
                    if (tt < ntap) quad[(co_l * 32 + l31) * G::KK + t0 + tt] = acc[tt][r];
            }
        }
        __syncthreads();
        const int cob = co0 + qm * 32, cib = ci0 + qn * 32;
        for (int e = tid; e < G::QUAD_FLOATS; e += G::NTHR) {
            const int co_l = e / (32 * G::KK), rem = e % (32 * G::KK);
            const int ci_l = rem / G::KK;
            if (cob + co_l < p.Cout && cib + ci_l < Cin) dst[((size_t)(cob + co_l) * Cin + cib) * G::KK + rem] = quad[e];
        }
    }
}

// out[i] = sum over the K-split slabs of part[k][i].  A block owns 32 elements; its eight groups of 32 lanes each sum an eighth
// of the slabs (in slab order, four loads in flight), the eight partial sums are added in group order: a fixed summation
// tree, so the result is bit-reproducible.  (One thread per element walking all slabs left a 64x64 weight gradient with
// 256 slabs at 1 TB/s: 64 us per launch.)
// (out1 / out2 / split: the fused q/k/v projection's gradient [3C][C] leaves as its three [C][C] tensors -- split = C * C)
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, int ksplit, size_t n,
                                                           float* __restrict__ out, float* __restrict__ out1 = nullptr,
                                                           float* __restrict__ out2 = nullptr, size_t split = 0) {
    __shared__ float red[8][32];
    const int seg = threadIdx.x >> 5, j = threadIdx.x & 31;
    const int per = (ksplit + 7) / 8;
    const int k0 = min(seg * per, ksplit), k1 = min(k0 + per, ksplit);
    for (size_t base = (size_t)blockIdx.x * 32; base < n; base += (size_t)gridDim.x * 32) {
        const size_t i = base + j;
        float s = 0.0f;
        if (i < n) {
            int k = k0;
            for (; k + 4 <= k1; k += 4) {
                const float a = part[(size_t)k * n + i], b = part[(size_t)(k + 1) * n + i];
                const float c = part[(size_t)(k + 2) * n + i], d = part[(size_t)(k + 3) * n + i];
                s += a; s += b; s += c; s += d;
            }
            for (; k < k1; ++k) s += part[(size_t)k * n + i];
        }
        red[seg][j] = s;
        __syncthreads();
        if (seg == 0 && i < n) {
            float t = red[0][j];
#pragma unroll
            for (int g = 1; g < 8; ++g) t += red[g][j];
            if (split == 0 || i < split) out[i] = t;
            else if (i < 2 * split) out1[i - split] = t;
            else out2[i - 2 * split] = t;
        }
        __syncthreads();
    }
}

// The same sums, element by element in the same order, FOUR consecutive elements per thread (16-byte loads and stores: the
// one-element form reads its slabs in 128-byte pieces); for n and split that are multiples of 4.
__global__ void __launch_bounds__(256) wgrad_reduce4_kernel(const float4* __restrict__ part, int ksplit, size_t n4,
                                                            float4* __restrict__ out, float4* __restrict__ out1,
                                                            float4* __restrict__ out2, size_t split4) {
    __shared__ float4 red[8][32];
    const int seg = threadIdx.x >> 5, j = threadIdx.x & 31;
    const int per = (ksplit + 7) / 8;
    const int k0 = min(seg * per, ksplit), k1 = min(k0 + per, ksplit);
    for (size_t base = (size_t)blockIdx.x * 32; base < n4; base += (size_t)gridDim.x * 32) {
        const size_t i = base + j;
        float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (i < n4) {
            int k = k0;
            for (; k + 4 <= k1; k += 4) {
                const float4 a = part[(size_t)k * n4 + i], b = part[(size_t)(k + 1) * n4 + i];
                const float4 c = part[(size_t)(k + 2) * n4 + i], d = part[(size_t)(k + 3) * n4 + i];
                s.x += a.x; s.x += b.x; s.x += c.x; s.x += d.x;
                s.y += a.y; s.y += b.y; s.y += c.y; s.y += d.y;
                s.z += a.z; s.z += b.z; s.z += c.z; s.z += d.z;
                s.w += a.w; s.w += b.w; s.w += c.w; s.w += d.w;
            }
            for (; k < k1; ++k) {
                const float4 a = part[(size_t)k * n4 + i];
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
        }
        red[seg][j] = s;
        __syncthreads();
        if (seg == 0 && i < n4) {
            float4 t = red[0][j];
#pragma unroll
            for (int g = 1; g < 8; ++g) { const float4 r = red[g][j]; t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w; }
            if (split4 == 0 || i < split4) out[i] = t;
            else if (i < 2 * split4) out1[i - split4] = t;
            else out2[i - 2 * split4] = t;
        }
        __syncthreads();
    }
}

template <int KS, int STRIDE, int TR, int TC>
static int launch_wgrad_cfg(sisic_ctx* ctx, WgradParams& p, hipStream_t s) {
    using G = WgradGeom<KS, STRIDE, TR, TC>;
    p.tiles_x = cdiv(p.Wout, TC);
    p.tiles_y = cdiv(p.Hout, TR);
    p.nblocks = p.B * p.tiles_x * p.tiles_y;
    p.blocks_per_slice = cdiv(p.nblocks, p.ksplit);
    p.ksplit = cdiv(p.nblocks, p.blocks_per_slice);       // no empty slices
    auto kern = conv_wgrad_kernel<KS, STRIDE, TR, TC>;
    static std::atomic<uint64_t> lds_opt_in{0};
    SISIC_TRY(ensure_dynamic_lds(ctx, reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_opt_in));
    const int64_t nwg = (int64_t)p.n_co_tiles * p.n_ci_tiles * p.ksplit;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(G::NTHR), G::LDS_BYTES, s, p);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

static int wgrad_ksplit(const WgradArgs& a, int Hout, int Wout) {
    // enough workgroups for 256 CUs: the weight tile count shrinks as the image grows and vice versa
    const int tiles = cdiv(a.Cout, 64) * cdiv(a.c0 + a.c1, 64);
    const int nblocks_min = a.B * cdiv(Hout, 8) * cdiv(Wout, 8);           // 64-pixel blocks
    // one workgroup per CU: measured level with two (26.8 vs 26.9 ms per step at batch 32, 64x64) and ahead at the
    // reference's batch 2, 128x128 (25.9 vs 26.9) -- half the partial slabs to write and re-read
    // (1x1: 16 accumulator registers, four workgroups fit a CU -- more, smaller slices: 61 -> 43 us per launch)
    return std::max(1, std::min(nblocks_min, cdiv(a.ksize == 1 ? 1024 : 256, tiles)));
}

#include "wgrad_winograd.inc"

static void wgrad_out_dims(const WgradArgs& a, int* Hout, int* Wout) {
    const int Hc = a.Hin << (a.ups ? 1 : 0), Wc = a.Win << (a.ups ? 1 : 0), pad = a.ksize / 2;
    *Hout = (Hc + 2 * pad - a.ksize) / a.stride + 1;
    *Wout = (Wc + 2 * pad - a.ksize) / a.stride + 1;
}

size_t conv_wgrad_scratch_floats(const WgradArgs& a) {
    int Hout, Wout;
    wgrad_out_dims(a, &Hout, &Wout);
    const size_t direct = (size_t)wgrad_ksplit(a, Hout, Wout) * a.Cout * (a.c0 + a.c1) * a.ksize * a.ksize;
    if (!wgrad_wino_applicable(a)) return direct;
    return std::max(direct, ((size_t)wgrad_wino_ksplit(a, Hout, Wout) + 1) * 16 * a.Cout * (a.c0 + a.c1));      // slabs + their sum
}

int launch_conv_wgrad(sisic_ctx* ctx, const WgradArgs& a, float* part, size_t part_floats, hipStream_t s) {
    SISIC_REQUIRE(a.in0 && a.dy && a.dw && part, "conv_wgrad: null tensor");
    SISIC_REQUIRE((a.ksize == 1 || a.ksize == 3) && (a.stride == 1 || a.stride == 2), "conv_wgrad: ksize %d stride %d", a.ksize, a.stride);
    SISIC_REQUIRE(!(a.ksize == 1 && (a.stride != 1 || a.ups)), "conv_wgrad: 1x1 is stride 1, no upsample");
    SISIC_REQUIRE((a.c1 == 0) == (a.in1 == nullptr), "conv_wgrad: in1/c1 mismatch");
    WgradParams p{};
    p.in0 = a.in0; p.in1 = a.in1; p.c0 = a.c0; p.c1 = a.c1;
    p.B = a.B; p.Hin = a.Hin; p.Win = a.Win; p.ups = a.ups ? 1 : 0;
    p.Hc = a.Hin << p.ups; p.Wc = a.Win << p.ups;
    wgrad_out_dims(a, &p.Hout, &p.Wout);
    p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.gn_silu = a.gn_silu;
    p.dy = a.dy; p.Cout = a.Cout; p.part = part;
    const int Cin = a.c0 + a.c1;
    p.n_co_tiles = cdiv(a.Cout, 64); p.n_ci_tiles = cdiv(Cin, 64);
    p.ksplit = wgrad_ksplit(a, p.Hout, p.Wout);
    const size_t n = (size_t)a.Cout * Cin * a.ksize * a.ksize;
    SISIC_REQUIRE((size_t)p.ksplit * n <= part_floats, "conv_wgrad: scratch too small");
    const double flops = 2.0 * a.B * a.Cout * (double)p.Hout * p.Wout * Cin * a.ksize * a.ksize;
    static const bool wino_on = [] { const char* e = std::getenv("SISIC_WGRAD_WINOGRAD"); return !e || std::atoi(e) != 0; }();
    {
        ProfileScope prof(ctx, s, PK_OTHER, 4.0 * a.B * ((double)Cin * a.Hin * a.Win + (double)a.Cout * p.Hout * p.Wout) + 4.0 * n, flops);
        if (a.ksize == 1) {
            // flat pixel rows: the image is one row of H*W pixels
            p.Hin = 1; p.Win = a.Hin * a.Win; p.Hc = 1; p.Wc = p.Win; p.Hout = 1; p.Wout = p.Win;
            SISIC_TRY((launch_wgrad_cfg<1, 1, 1, 64>(ctx, p, s)));
        } else if (a.stride == 1 && wino_on) {
            // Winograd-domain form (wgrad_winograd.inc): its own reduction, so it returns here
            SISIC_REQUIRE(((size_t)wgrad_wino_ksplit(a, p.Hout, p.Wout) + 1) * 16 * a.Cout * Cin <= part_floats, "conv_wgrad: scratch too small");
            return launch_wgrad_wino(ctx, p, a, s);
        } else if (a.stride == 2) {
            if (p.Wout > 8) SISIC_TRY((launch_wgrad_cfg<3, 2, 2, 16>(ctx, p, s)));
            else SISIC_TRY((launch_wgrad_cfg<3, 2, 4, 8>(ctx, p, s)));
        } else if (p.Wout > 8) {        // (2 x 32 pixel blocks for wide images were measured: 34-float halo rows, 21 spilled registers, 3 % slower)
            SISIC_TRY((launch_wgrad_cfg<3, 1, 4, 16>(ctx, p, s)));
        } else {
            SISIC_TRY((launch_wgrad_cfg<3, 1, 8, 8>(ctx, p, s)));
        }
    }
    const size_t split = a.dw1 ? n / 3 : (size_t)0;
    const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if ((n & 3) == 0 && (split & 3) == 0 && al16(part) && al16(a.dw) && al16(a.dw1) && al16(a.dw2)) {
        const size_t n4 = n >> 2;
        const int blocks = (int)std::min<size_t>((n4 + 31) / 32, 8192);
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(part), p.ksplit, n4,
                           reinterpret_cast<float4*>(a.dw), reinterpret_cast<float4*>(a.dw1), reinterpret_cast<float4*>(a.dw2), split >> 2);
        SISIC_HIP(hipGetLastError());
        return SISIC_OK;
    }
    const int blocks = (int)std::min<size_t>((n + 31) / 32, 8192);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, part, p.ksplit, n, a.dw, a.dw1, a.dw2, split);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// W'[ci][co][KK-1-t] = W[co][ci][t]: the filter of the backward-data convolution
__global__ void transpose_flip_kernel(const float* __restrict__ w, int Cout, int Cin, int KK, float* __restrict__ wt) {
    const size_t n = (size_t)Cout * Cin * KK;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        transpose_flip_elem(i, w, Cout, Cin, KK, wt);
    }
}

int launch_transpose_flip(sisic_ctx*, const float* w, int Cout, int Cin, int KK, float* wt, hipStream_t s) {
    const size_t n = (size_t)Cout * Cin * KK;
    hipLaunchKernelGGL(transpose_flip_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, s, w, Cout,
                       Cin, KK, wt);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// ================================================================ reductions over planes / rows =====================
// out[plane] = sum of the plane's HW values; one wave per plane
__global__ void __launch_bounds__(256) plane_sum_kernel(const float* __restrict__ x, int planes, int HW, float* __restrict__ out) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (plane >= planes) return;
    const float* src = x + (size_t)plane * HW;
    float s = 0.0f;
    for (int i = lane; i < HW; i += 64) s += src[i];
    s = wave_sum_t(s);
    if (lane == 0) out[plane] = s;
}

int launch_plane_sums(sisic_ctx*, const float* x, int planes, int HW, float* out, hipStream_t s) {
    hipLaunchKernelGGL(plane_sum_kernel, dim3(cdiv(planes, 4)), dim3(256), 0, s, x, planes, HW, out);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// out[c] (+)= sum_r m[r * ld + c]
__global__ void col_sum_kernel(const float* __restrict__ m, int rows, int cols, int ld, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.0f;
    for (int r = 0; r < rows; ++r) s += m[(size_t)r * ld + c];
    out[c] = accumulate ? out[c] + s : s;
}

// two matrices at once (GroupNorm backward: dgamma and dbeta over the batch): same per-column order as col_sum_kernel
__global__ void col_sum2_kernel(const float* __restrict__ ma, const float* __restrict__ mb, int rows, int cols,
                                float* __restrict__ outa, float* __restrict__ outb) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float sa = 0.0f, sb = 0.0f;
    for (int r = 0; r < rows; ++r) {
        sa += ma[(size_t)r * cols + c];
        sb += mb[(size_t)r * cols + c];
    }
    outa[c] = sa;
    outb[c] = sb;
}

int launch_col_sums(sisic_ctx*, const float* m, int rows, int cols, int ld, float* out, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(col_sum_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, s, m, rows, cols, ld, out, accumulate);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// Bias gradient of one convolution in ONE launch (was plane_sum_kernel + col_sum_kernel (+ copy_cols_kernel, + three device
// copies for the fused q/k/v projection): ~200 launches of 5-10 us per training step).  One workgroup per output channel:
// its sixteen waves sum the channel's B planes (plane b by wave b % 16, lanes striding the pixels, then the wave butterfly --
// the order of plane_sum_kernel), lane 0 of each wave leaves S[b] in LDS; then S[b] goes to the time-embedding gradient
// (column tproj_col + c of a [B, tproj_ld] matrix) when the convolution added a projected embedding, and the sum over b in
// batch order (the order of col_sum_kernel) is the bias gradient.  The fused q/k/v projection writes its three biases.
// (Until round 3 the same sums in the same order as the kernels it replaced; the 16-byte loads sum a plane in another order.)
// (sixteen waves per channel: with four, a 64-channel layer at batch 32 was 64 blocks walking 8 planes each -- 47 us)
__global__ void __launch_bounds__(1024) bias_grad_kernel(const float* __restrict__ dy, int B, int C, int HW, float* __restrict__ db0,
                                                         float* __restrict__ db1, float* __restrict__ db2, int split,
                                                         float* __restrict__ tproj, int tproj_ld) {
    extern __shared__ float S[];          // [B]
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = wave; b < B; b += 16) {
        const float* src = dy + ((size_t)b * C + c) * HW;
        float s = 0.0f;
        if ((HW & 3) == 0) {          // 16 bytes per lane and load, four running sums (round 3: a quarter of the load instructions)
            const float4* src4 = reinterpret_cast<const float4*>(src);
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            for (int i = lane; i < (HW >> 2); i += 64) {
                const float4 v = src4[i];
                s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
            }
            s = (s0 + s1) + (s2 + s3);
        } else {
            for (int i = lane; i < HW; i += 64) s += src[i];
        }
        s = wave_sum_t(s);
        if (lane == 0) S[b] = s;
    }
    __syncthreads();
    if (tproj)
        for (int b = threadIdx.x; b < B; b += 1024) tproj[(size_t)b * tproj_ld + c] = S[b];
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int b = 0; b < B; ++b) t += S[b];
        if (split <= 0) db0[c] = t;
        else if (c < split) db0[c] = t;
        else if (c < 2 * split) db1[c - split] = t;
        else db2[c - 2 * split] = t;
    }
}

int launch_bias_grad(sisic_ctx*, const float* dy, int B, int C, int HW, float* db0, float* db1, float* db2, int split, float* tproj,
                     int tproj_ld, hipStream_t s) {
    hipLaunchKernelGGL(bias_grad_kernel, dim3(C), dim3(1024), (size_t)B * sizeof(float), s, dy, B, C, HW, db0, db1, db2, split, tproj,
                       tproj_ld);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// dst[r * ld_dst + c] = src[r * cols + c]
__global__ void copy_cols_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst, int ld_dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    dst[(size_t)(i / cols) * ld_dst + i % cols] = src[i];
}

int launch_copy_cols(sisic_ctx*, const float* src, int rows, int cols, float* dst, int ld_dst, hipStream_t s) {
    hipLaunchKernelGGL(copy_cols_kernel, dim3(cdiv(rows * cols, 256)), dim3(256), 0, s, src, rows, cols, dst, ld_dst);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// ================================================================ GroupNorm (+SiLU) backward ========================
// Forward: u = x * scale[b,c] + shift[b,c] (= gamma xhat + beta), a = silu(u) or u.  Given da:
//   du = da * silu'(u);  dgamma_c = sum_{b,hw} du xhat;  dbeta_c = sum_{b,hw} du
//   dx = rstd * (gamma_c du - mean_g(gamma du) - xhat * mean_g(gamma du xhat))        (means over the group's cpg*HW elements)
// Pass 1: per (b,c) plane the two sums A = sum du, Bx = sum du xhat.   Pass 2: dx, ADDED to the gradient of x.
__global__ void __launch_bounds__(256)
gn_bwd_reduce_kernel(const float* __restrict__ da, const float* __restrict__ in0, int c0, const float* __restrict__ in1, int c1,
                     int HW, int groups, const float* __restrict__ scale, const float* __restrict__ shift,
                     const float* __restrict__ mean_rstd, int silu, float* __restrict__ sumA, float* __restrict__ sumB) {
    __shared__ float red[4];
    const int C = c0 + c1, cpg = C / groups;
    const int b = blockIdx.x / C, c = blockIdx.x % C, g = c / cpg;
    const float* x = c < c0 ? in0 + ((size_t)b * c0 + c) * HW : in1 + ((size_t)b * c1 + (c - c0)) * HW;
    const float* d = da + ((size_t)b * C + c) * HW;
    const float sc = scale[(size_t)b * C + c], sh = shift[(size_t)b * C + c];
    const float mean = mean_rstd[2 * ((size_t)b * groups + g)], rstd = mean_rstd[2 * ((size_t)b * groups + g) + 1];
    float a = 0.0f, bx = 0.0f;
    if ((HW & 3) == 0) {              // 16 bytes per lane and load (round 3); a thread's four elements in index order
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* d4 = reinterpret_cast<const float4*>(d);
        for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
            const float4 xv = x4[i], dv = d4[i];
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            const float ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float du = ds[k];
                if (silu) du *= silu_grad(xs[k] * sc + sh);
                a += du;
                bx += du * ((xs[k] - mean) * rstd);
            }
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) {
            const float xv = x[i];
            float du = d[i];
            if (silu) du *= silu_grad(xv * sc + sh);
            a += du;
            bx += du * ((xv - mean) * rstd);
        }
    }
    a = block_sum_256(a, red);
    bx = block_sum_256(bx, red);
    if (threadIdx.x == 0) {
        sumA[(size_t)b * C + c] = a;
        sumB[(size_t)b * C + c] = bx;
    }
}

__global__ void __launch_bounds__(256)
gn_bwd_apply_kernel(const float* __restrict__ da, const float* __restrict__ in0, int c0, const float* __restrict__ in1, int c1,
                    int HW, int groups, const float* __restrict__ scale, const float* __restrict__ shift,
                    const float* __restrict__ mean_rstd, const float* __restrict__ gamma, int silu,
                    const float* __restrict__ sumA, const float* __restrict__ sumB, float* __restrict__ g0,
                    float* __restrict__ g1, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int C = c0 + c1, cpg = C / groups;
    const int b = blockIdx.x / C, c = blockIdx.x % C, g = c / cpg;
    // the parameter gradients ride along (they were a launch of their own, col_sum2_kernel: one to three workgroups walking the
    // batch, 6.6 us x 51 norms): the channel's first block sums its column over the batch, in batch order as before
    if (b == 0 && threadIdx.x == 0) {
        const int Bn = gridDim.x / C;
        float sg = 0.0f, sb = 0.0f;
        for (int r = 0; r < Bn; ++r) {
            sg += sumB[(size_t)r * C + c];
            sb += sumA[(size_t)r * C + c];
        }
        dgamma[c] = sg;
        dbeta[c] = sb;
    }
    const float* x = c < c0 ? in0 + ((size_t)b * c0 + c) * HW : in1 + ((size_t)b * c1 + (c - c0)) * HW;
    float* dst = c < c0 ? g0 + ((size_t)b * c0 + c) * HW : g1 + ((size_t)b * c1 + (c - c0)) * HW;
    const float* d = da + ((size_t)b * C + c) * HW;
    const float sc = scale[(size_t)b * C + c], sh = shift[(size_t)b * C + c];
    const float mean = mean_rstd[2 * ((size_t)b * groups + g)], rstd = mean_rstd[2 * ((size_t)b * groups + g) + 1];
    float mA = 0.0f, mB = 0.0f;                 // every thread: the group's sums (cpg <= 16 values each), same order
    for (int j = 0; j < cpg; ++j) {
        const int cj = g * cpg + j;
        mA += gamma[cj] * sumA[(size_t)b * C + cj];
        mB += gamma[cj] * sumB[(size_t)b * C + cj];
    }
    const float inv_m = 1.0f / ((float)cpg * (float)HW);
    mA *= inv_m; mB *= inv_m;
    const float gm = gamma[c];
    if ((HW & 3) == 0) {              // 16 bytes per lane and access (round 3): the same arithmetic per element
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* d4 = reinterpret_cast<const float4*>(d);
        float4* dst4 = reinterpret_cast<float4*>(dst);
        for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
            const float4 xv = x4[i], dv = d4[i];
            float4 o = dst4[i];
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            const float ds[4] = {dv.x, dv.y, dv.z, dv.w};
            float os[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float du = ds[k];
                if (silu) du *= silu_grad(xs[k] * sc + sh);
                const float xh = (xs[k] - mean) * rstd;
                os[k] += rstd * (gm * du - mA - xh * mB);
            }
            dst4[i] = make_float4(os[0], os[1], os[2], os[3]);
        }
        return;
    }
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float xv = x[i];
        float du = d[i];
        if (silu) du *= silu_grad(xv * sc + sh);
        const float xh = (xv - mean) * rstd;
        dst[i] += rstd * (gm * du - mA - xh * mB);
    }
}

int launch_gn_bwd(sisic_ctx* ctx, const float* da, const float* in0, int c0, const float* in1, int c1, int B, int HW, int groups,
                  const float* scale, const float* shift, const float* mean_rstd, const float* gamma, int silu,
                  float* sums, float* g0, float* g1, float* dgamma, float* dbeta, hipStream_t s) {
    const int C = c0 + c1;
    SISIC_REQUIRE(da && in0 && scale && shift && mean_rstd && gamma && sums && g0 && (c1 == 0 || (in1 && g1)), "gn_bwd: null tensor");
    float* sumA = sums;
    float* sumB = sums + (size_t)B * C;
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(B * C), dim3(256), 0, s, da, in0, c0, in1, c1, HW, groups, scale, shift,
                       mean_rstd, silu, sumA, sumB);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(B * C), dim3(256), 0, s, da, in0, c0, in1, c1, HW, groups, scale, shift, mean_rstd,
                       gamma, silu, sumA, sumB, g0, g1, dgamma, dbeta);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// ================================================================ gradient routing ==================================
// g0[b, c, :] += da[b, c, :] (c < c0);  g1[b, c - c0, :] += da[b, c, :]
__global__ void accum_split_kernel(const float* __restrict__ da, int C, int HW, float* __restrict__ g0, int c0,
                                   float* __restrict__ g1, int c1, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % HW);
        const size_t bc = i / HW;
        const int c = (int)(bc % C);
        const size_t b = bc / C;
        if (c < c0) g0[(b * c0 + c) * HW + px] += da[i];
        else g1[(b * c1 + (c - c0)) * HW + px] += da[i];
    }
}

// (16 bytes per lane: four pixels of one plane -- a quarter of the index arithmetic and of the memory instructions)
__global__ void accum_split4_kernel(const float4* __restrict__ da, int C, int HW4, float4* __restrict__ g0, int c0,
                                    float4* __restrict__ g1, int c1, size_t total4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % HW4);
        const size_t bc = i / HW4;
        const int c = (int)(bc % C);
        const size_t b = bc / C;
        float4* dst = c < c0 ? g0 + (b * c0 + c) * HW4 + px : g1 + (b * c1 + (c - c0)) * HW4 + px;
        const float4 v = da[i];
        float4 o = *dst;
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
        *dst = o;
    }
}

int launch_accum_split(sisic_ctx*, const float* da, int B, int C, int HW, float* g0, int c0, float* g1, int c1, hipStream_t s) {
    const size_t total = (size_t)B * C * HW;
    if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(da) | reinterpret_cast<uintptr_t>(g0) | reinterpret_cast<uintptr_t>(g1)) & 15) == 0) {
        const size_t total4 = total >> 2;
        hipLaunchKernelGGL(accum_split4_kernel, dim3((unsigned)std::min<size_t>((total4 + 255) / 256, 4096)), dim3(256), 0, s,
                           reinterpret_cast<const float4*>(da), C, HW >> 2, reinterpret_cast<float4*>(g0), c0, reinterpret_cast<float4*>(g1), c1, total4);
        SISIC_HIP(hipGetLastError());
        return SISIC_OK;
    }
    hipLaunchKernelGGL(accum_split_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, da, C, HW,
                       g0, c0, g1, c1, total);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// nearest-2x upsample backward: g[b,c,y,x] += sum of the 2x2 block of da[b,c,2y..,2x..]
__global__ void accum_pool2_kernel(const float* __restrict__ da, int H, int W, float* __restrict__ g, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const size_t plane = i / ((size_t)W * H);
        const float* src = da + plane * 4 * H * W + (size_t)(2 * y) * 2 * W + 2 * x;
        g[i] += (src[0] + src[1]) + (src[2 * W] + src[2 * W + 1]);
    }
}

int launch_accum_pool2(sisic_ctx*, const float* da, int planes, int H, int W, float* g, hipStream_t s) {
    const size_t total = (size_t)planes * H * W;
    hipLaunchKernelGGL(accum_pool2_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, da, H, W, g,
                       total);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

__global__ void add_inplace4_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        float4 o = dst[i];
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
        dst[i] = o;
    }
}

int launch_add_inplace(sisic_ctx*, float* dst, const float* src, size_t n, hipStream_t s) {
    if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0) {
        const size_t n4 = n >> 2;
        hipLaunchKernelGGL(add_inplace4_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096)), dim3(256), 0, s,
                           reinterpret_cast<float4*>(dst), reinterpret_cast<const float4*>(src), n4);
        SISIC_HIP(hipGetLastError());
        return SISIC_OK;
    }
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, dst, src, n);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// ================================================================ attention backward ================================
// One workgroup per (sample, head), d = 8.  q,k,v,dO of the head in LDS ([N][8] rows).  Softmax is recomputed:
//   pass A (thread = query i):  m_i, l_i (online), D_i = dO_i . O_i,  dQ_i = scale * sum_j P_ij (dO_i . v_j - D_i) k_j
//   pass B (thread = key j):    dV_j = sum_i P_ij dO_i,  dK_j = scale * sum_i P_ij (dO_i . v_j - D_i) q_i
// P_ij = exp(scale q_i . k_j - m_i) / l_i.  No atomics: every output row has one owner.  exp = v_exp_f32 (__expf), as in the
// forward kernel: the accurate expf is ~15 vector instructions of the ~40 per (query, key) pair and pass.
constexpr int ATB_THREADS = 256;

__global__ void __launch_bounds__(ATB_THREADS)
attention_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ dO,
                     float* __restrict__ dqkv, int C, int N, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const qs = smem;                 // [N][8]
    float* const ks_ = qs + (size_t)N * 8;
    float* const vs = ks_ + (size_t)N * 8;
    float* const gs = vs + (size_t)N * 8;   // dO
    float* const ms = gs + (size_t)N * 8;   // [N] row max
    float* const ls = ms + N;               // [N] 1 / row sum
    float* const ds = ls + N;               // [N] D_i
    const int heads = C / 8;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const size_t qoff = ((size_t)b * 3 * C + h * 8) * N, koff = qoff + (size_t)C * N, voff = koff + (size_t)C * N;
    const size_t ooff = ((size_t)b * C + h * 8) * N;
    for (int i = threadIdx.x; i < 8 * N; i += ATB_THREADS) {
        const int d = i / N, n = i % N;                       // global reads run along n (contiguous)
        qs[n * 8 + d] = qkv[qoff + (size_t)d * N + n];
        ks_[n * 8 + d] = qkv[koff + (size_t)d * N + n];
        vs[n * 8 + d] = qkv[voff + (size_t)d * N + n];
        gs[n * 8 + d] = dO[ooff + (size_t)d * N + n];
    }
    __syncthreads();
    // ---- pass A
    for (int i = threadIdx.x; i < N; i += ATB_THREADS) {
        float q[8], g[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) { q[d] = qs[i * 8 + d] * scale; g[d] = gs[i * 8 + d]; }
        float m = -INFINITY, l = 0.0f;
        for (int j = 0; j < N; ++j) {
            float sc = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) sc += q[d] * ks_[j * 8 + d];
            const float mn = fmaxf(m, sc);
            l = l * __expf(m - mn) + __expf(sc - mn);
            m = mn;
        }
        float D = 0.0f;
#pragma unroll
        for (int d = 0; d < 8; ++d) D += g[d] * o[ooff + (size_t)d * N + i];
        const float il = 1.0f / l;
        float dq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < N; ++j) {
            float sc = 0.0f, dp = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) { sc += q[d] * ks_[j * 8 + d]; dp += g[d] * vs[j * 8 + d]; }
            const float dsij = __expf(sc - m) * il * (dp - D);
#pragma unroll
            for (int d = 0; d < 8; ++d) dq[d] += dsij * ks_[j * 8 + d];
        }
        ms[i] = m; ls[i] = il; ds[i] = D;
#pragma unroll
        for (int d = 0; d < 8; ++d) dqkv[qoff + (size_t)d * N + i] = dq[d] * scale;
    }
    __syncthreads();
    // ---- pass B
    for (int j = threadIdx.x; j < N; j += ATB_THREADS) {
        float k[8], v[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) { k[d] = ks_[j * 8 + d] * scale; v[d] = vs[j * 8 + d]; }
        float dk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < N; ++i) {
            float sc = 0.0f, dp = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) { sc += qs[i * 8 + d] * k[d]; dp += gs[i * 8 + d] * v[d]; }
            const float pij = __expf(sc - ms[i]) * ls[i];
            const float dsij = pij * (dp - ds[i]);
#pragma unroll
            for (int d = 0; d < 8; ++d) { dv[d] += pij * gs[i * 8 + d]; dk[d] += dsij * qs[i * 8 + d]; }
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            dqkv[koff + (size_t)d * N + j] = dk[d] * scale;
            dqkv[voff + (size_t)d * N + j] = dv[d];
        }
    }
}

int launch_attention_bwd(sisic_ctx* ctx, const float* qkv, const float* o, const float* dO, float* dqkv, int B, int C, int N,
                         int head_dim, hipStream_t s) {
    SISIC_REQUIRE(qkv && o && dO && dqkv && head_dim == 8 && C % 8 == 0, "attention_bwd: bad arguments");
    const size_t lds = ((size_t)N * 32 + 3 * (size_t)N) * sizeof(float);
    SISIC_REQUIRE(lds <= 160 * 1024, "attention_bwd: %d tokens do not fit the LDS (max 1170)", N);
    static std::atomic<uint64_t> lds_opt_in{0};
    SISIC_TRY(ensure_dynamic_lds(ctx, reinterpret_cast<const void*>(attention_bwd_kernel), 160 * 1024, lds_opt_in));
    ProfileScope prof(ctx, s, PK_ATTN, 32.0 * B * C * N, 10.0 * B * C * double(N) * N);
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(B * (C / 8)), dim3(ATB_THREADS), lds, s, qkv, o, dO, dqkv, C, N, 0.35355339059327373f);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// ================================================================ small linears (time embedding) ====================
// dW[r][k] = sum_b dy[b*ld + r] * x[b*K + k]
__global__ void linear_wgrad_kernel(const float* __restrict__ dy, int ld, const float* __restrict__ x, int B, int R, int K,
                                    float* __restrict__ dW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * K) return;
    const int r = (int)(i / K), k = (int)(i % K);
    float s = 0.0f;
    for (int b = 0; b < B; ++b) s += dy[(size_t)b * ld + r] * x[(size_t)b * K + k];
    dW[i] = s;
}

int launch_linear_wgrad(sisic_ctx*, const float* dy, int ld, const float* x, int B, int R, int K, float* dW, hipStream_t s) {
    hipLaunchKernelGGL(linear_wgrad_kernel, dim3((unsigned)(((size_t)R * K + 255) / 256)), dim3(256), 0, s, dy, ld, x, B, R, K, dW);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// dx[b][k] = sum_r dy[b*ld + r] * W[r][k];  W is [R][K] row-major, or (transposed) stored as [K][R]
__global__ void linear_dgrad_kernel(const float* __restrict__ dy, int ld, const float* __restrict__ W, int R, int K,
                                    float* __restrict__ dx, int transposed) {
    const int b = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const size_t sr = transposed ? 1 : (size_t)K, sk = transposed ? (size_t)R : 1;
    float s = 0.0f;
    for (int r = 0; r < R; ++r) s += dy[(size_t)b * ld + r] * W[r * sr + k * sk];
    dx[(size_t)b * K + k] = s;
}

// the same for a weight stored transposed ([K][R]): one wave per (b, k), lanes sweep r -- both operands stream contiguously
// (a thread per (b, k) would walk rows R floats apart: 306 us for the 4032-row time-embedding projection against ~10)
__global__ void __launch_bounds__(256) linear_dgrad_t_kernel(const float* __restrict__ dy, int ld, const float* __restrict__ Wt, int R,
                                                             int K, int total, float* __restrict__ dx) {
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= total) return;
    const int b = item / K, k = item % K;
    const float* d = dy + (size_t)b * ld;
    const float* w = Wt + (size_t)k * R;
    float s = 0.0f;
    for (int r = lane; r < R; r += 64) s += d[r] * w[r];
    s = wave_sum_t(s);
    if (lane == 0) dx[item] = s;
}

int launch_linear_dgrad(sisic_ctx*, const float* dy, int ld, const float* W, int B, int R, int K, float* dx, hipStream_t s,
                        int w_is_transposed) {
    if (w_is_transposed) {
        hipLaunchKernelGGL(linear_dgrad_t_kernel, dim3(cdiv(B * K, 4)), dim3(256), 0, s, dy, ld, W, R, K, B * K, dx);
        SISIC_HIP(hipGetLastError());
        return SISIC_OK;
    }
    hipLaunchKernelGGL(linear_dgrad_kernel, dim3(cdiv(K, 256), B), dim3(256), 0, s, dy, ld, W, R, K, dx, w_is_transposed);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// out = silu(pre) with the exact division (the time-embedding MLP's form, elementwise.hip)
__global__ void silu_fwd_kernel(const float* __restrict__ pre, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pre[i] / (1.0f + expf(-pre[i]));
}

int launch_silu_fwd(sisic_ctx*, const float* pre, size_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(silu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pre, n, out);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// out = dy * silu'(pre)
__global__ void silu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = dy[i] * silu_grad(pre[i]);
}

int launch_silu_bwd(sisic_ctx*, const float* dy, const float* pre, size_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dy, pre, n, out);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// ================================================================ loss, optimizer ===================================
// loss = mean((pred - target)^2)  (F.mse_loss, train_diffusion.py:219);  dpred = grad_scale * 2 (pred - target) / n
__global__ void __launch_bounds__(256) mse_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target, size_t n,
                                                          float gscale, float* __restrict__ dpred, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = pred[i] - target[i];
        s += d * d;
        if (dpred) dpred[i] = gscale * d;
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) mse_final_kernel(const float* __restrict__ part, int nparts, float inv_n, float* __restrict__ loss) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) loss[0] = s * inv_n;
}

int launch_mse(sisic_ctx*, const float* pred, const float* target, size_t n, float grad_scale, float* loss_dev, float* dpred,
               float* part, int nparts, hipStream_t s) {
    SISIC_REQUIRE(pred && target && loss_dev && part && n > 0 && nparts > 0, "mse: bad arguments");
    const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)nparts);
    hipLaunchKernelGGL(mse_partial_kernel, dim3(blocks), dim3(256), 0, s, pred, target, n, grad_scale * 2.0f / (float)n, dpred, part);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, part, blocks, 1.0f / (float)n, loss_dev);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// flag |= 1 when any gradient is inf / nan (GradScaler's found_inf)
__global__ void check_finite_kernel(const float* __restrict__ g, size_t n, int* __restrict__ flag) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        bad = bad || !isfinite(g[i]);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

int launch_check_finite(sisic_ctx*, const float* g, size_t n, int* flag, hipStream_t s) {
    hipLaunchKernelGGL(check_finite_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, s, g, n, flag);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// torch.optim.Adam (single-tensor form, no weight decay, no amsgrad), in its operation order:
//   m = lerp(m, g, 1 - b1);  v = v * b2 + (1 - b2) g g;  denom = sqrt(v) / sqrt(bc2) + eps;  p = p - (lr / bc1) * m / denom
// g is first multiplied by inv_scale (GradScaler.unscale_).
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float one_minus_b1, float b2, float one_minus_b2, float step_size, float bc2_sqrt, float eps,
                            float inv_scale) {
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * inv_scale;
        const float mi = m[i] + one_minus_b1 * (gi - m[i]);
        const float vi = v[i] * b2 + one_minus_b2 * (gi * gi);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

int launch_adam(sisic_ctx*, float* p, const float* g, float* m, float* v, size_t n, double lr, double b1, double b2, double eps,
                int64_t step, float inv_scale, hipStream_t s) {
    // the scalars as torch forms them: Python doubles (1 - beta, lr / bias_correction1, sqrt(bias_correction2)) rounded to
    // fp32 once, when they meet the fp32 tensors
    const double bc1 = 1.0 - std::pow(b1, (double)step), bc2 = 1.0 - std::pow(b2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, p, g, m, v, n,
                       (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)(lr / bc1), (float)std::sqrt(bc2), (float)eps,
                       inv_scale);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// DDPMScheduler.add_noise (train_diffusion.py:217):  out[b] = a[b] * x0[b] + c[b] * noise[b]   (two products, one sum, fp32)
__global__ void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const float* __restrict__ a,
                                 const float* __restrict__ c, float* __restrict__ out, size_t per, size_t total) {
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per;
        const float t0 = a[b] * x0[i];
        const float t1 = c[b] * noise[i];
        out[i] = t0 + t1;
    }
}

int launch_add_noise(sisic_ctx*, const float* x0, const float* noise, const float* a_dev, const float* c_dev, float* out, int B,
                     size_t per, hipStream_t s) {
    const size_t total = (size_t)B * per;
    hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, x0, noise,
                       a_dev, c_dev, out, per, total);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

}  // namespace sisic
