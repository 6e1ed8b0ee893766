// groupnorm.hip -- GroupNorm statistics, folded with the affine parameters.
//
// Replaces the reduction half of torch.nn.GroupNorm(32, C, eps=1e-5) in ResnetBlock2D
// (norm1/norm2), Attention.group_norm and conv_norm_out.
// The normalise+affine(+SiLU) half is fused into the consuming convolution's load path
// (conv_mfma.hip), so the activation tensor is read once here and once by the conv.
//
// One workgroup per (sample, group): one pass over the group's cpg*HW elements accumulating
// shifted first and second moments, wavefront-shuffle + LDS reductions, then
//   scale[b,c] = gamma[c]*rstd,  shift[b,c] = beta[c] - mean*scale[b,c].
// The input may be the channel concatenation of two tensors (up-path skip connections);
// a group may straddle the seam.
//
// HBM-bound.  Algorithmic bytes per launch: 4*B*C*HW (read once) + 8*B*C (written).
#include "common.h"
#include "gn_merge.h"

namespace sisic {

constexpr int GN_THREADS = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();   // protect red[] from the previous use
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < GN_THREADS / 64; ++w) t += red[w];
    return t;
}

__global__ void __launch_bounds__(GN_THREADS)
gn_stats_kernel(const float* __restrict__ in0, int c0, const float* __restrict__ in1, int c1, int HW, int groups,
                float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                float* __restrict__ scale, float* __restrict__ shift, int aligned16, float* __restrict__ mean_rstd) {
    __shared__ float red[GN_THREADS / 64];
    const int C = c0 + c1;
    const int cpg = C / groups;
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int tid = threadIdx.x;
    const bool vec = aligned16 && (HW & 3) == 0;

    auto plane = [&](int c) -> const float* {
        return (c < c0) ? in0 + ((size_t)b * c0 + c) * HW : in1 + ((size_t)b * c1 + (c - c0)) * HW;
    };

    // ONE pass over the data with shifted sums: K = the group's first element, s1 = sum(x-K),
    // s2 = sum((x-K)^2); mean = K + s1/n, var = s2/n - (s1/n)^2.  Shifting by a sample of the data keeps the
    // cancellation in the variance benign (|mean-K| is O(std)), at half the memory traffic of two passes.
    const float K = plane(g * cpg)[0];
    float s1 = 0.0f, s2 = 0.0f;
    for (int j = 0; j < cpg; ++j) {
        const float* src = plane(g * cpg + j);
        if (vec) {
            const float4* s4 = reinterpret_cast<const float4*>(src);
            const int n4 = HW / 4;
            int i = tid;
            // four independent 16-byte loads in flight per thread: the kernel is latency/bandwidth-bound
            for (; i + 3 * GN_THREADS < n4; i += 4 * GN_THREADS) {
                const float4 v0 = s4[i], v1 = s4[i + GN_THREADS], v2 = s4[i + 2 * GN_THREADS], v3 = s4[i + 3 * GN_THREADS];
                const float4 vv[4] = {v0, v1, v2, v3};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float a = vv[u].x - K, bq = vv[u].y - K, c = vv[u].z - K, d = vv[u].w - K;
                    s1 += (a + bq) + (c + d);
                    s2 += (a * a + bq * bq) + (c * c + d * d);
                }
            }
            for (; i < n4; i += GN_THREADS) {
                const float4 v = s4[i];
                const float a = v.x - K, bq = v.y - K, c = v.z - K, d = v.w - K;
                s1 += (a + bq) + (c + d);
                s2 += (a * a + bq * bq) + (c * c + d * d);
            }
        } else {
            for (int i = tid; i < HW; i += GN_THREADS) {
                const float a = src[i] - K;
                s1 += a;
                s2 += a * a;
            }
        }
    }
    const float n = (float)cpg * (float)HW;
    const float m1 = block_sum(s1, red) / n;
    const float m2 = block_sum(s2, red) / n;
    const float mean = K + m1;
    const float var = fmaxf(m2 - m1 * m1, 0.0f);
    const float rstd = 1.0f / sqrtf(var + eps);
    if (mean_rstd && tid == 0) {
        mean_rstd[2 * (size_t)blockIdx.x] = mean;
        mean_rstd[2 * (size_t)blockIdx.x + 1] = rstd;
    }
    if (tid < cpg) {
        const int c = g * cpg + tid;
        const float sc = gamma[c] * rstd;
        scale[(size_t)b * C + c] = sc;
        shift[(size_t)b * C + c] = beta[c] - mean * sc;
    }
}

int launch_gn_stats(sisic_ctx* ctx, const float* in0, int c0, const float* in1, int c1, int B, int HW, int groups,
                    float eps, const float* gamma, const float* beta, float* scale, float* shift, hipStream_t s,
                    float* mean_rstd) {
    SISIC_REQUIRE(in0 && gamma && beta && scale && shift, "groupnorm_stats: null tensor");
    SISIC_REQUIRE((c1 == 0) == (in1 == nullptr), "groupnorm_stats: in1/c1 mismatch");
    const int C = c0 + c1;
    SISIC_REQUIRE(B > 0 && HW > 0 && groups > 0 && C % groups == 0, "groupnorm_stats: C=%d not divisible by groups=%d", C, groups);
    SISIC_REQUIRE(C / groups <= GN_THREADS, "groupnorm_stats: %d channels per group unsupported", C / groups);
    ProfileScope prof(ctx, s, PK_GN, 4.0 * B * C * HW + 8.0 * B * C, 0.0);
    const int aligned16 = ((reinterpret_cast<uintptr_t>(in0) | reinterpret_cast<uintptr_t>(in1)) & 15) == 0;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(B * groups), dim3(GN_THREADS), 0, s, in0, c0, in1, c1, HW, groups, eps,
                       gamma, beta, scale, shift, aligned16, mean_rstd);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// Sum of a double over the 64 lanes in a fixed order: DPP butterflies on the two 32-bit halves inside each row of 16
// lanes (a few cycles each; ds_bpermute shuffles cost an LDS round trip per step), then the four row totals by readlane.
__device__ __forceinline__ double dpp_f64(double v, const int ctrl_sel) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    switch (ctrl_sel) {
        case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true); break;
        case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, true); break;
        case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xf, 0xf, true); break;
        default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xf, 0xf, true); break;
    }
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave64_sum_f64(double v) {
    v += dpp_f64(v, 0);      // quad_perm [1,0,3,2]
    v += dpp_f64(v, 1);      // quad_perm [2,3,0,1]
    v += dpp_f64(v, 2);      // row_half_mirror
    v += dpp_f64(v, 3);      // row_mirror
    const long long b = __double_as_longlong(v);
    const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    double r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        r[i] = __longlong_as_double(((long long)__builtin_amdgcn_readlane(hi, 16 * i) << 32) |
                                    (unsigned int)__builtin_amdgcn_readlane(lo, 16 * i));
    return (r[0] + r[1]) + (r[2] + r[3]);
}

// Finalize from per-workgroup partials (count, sum, M2 about the partial's own mean): one wave per (image, group).
// Lanes stride over the (channel, slot) pairs of the group; the pairwise-merge identity
//     M2 = sum_i M2_i + sum_i n_i (mean_i - mean)^2
// is evaluated in float64 with fixed-order butterflies, so the result is independent of timing and as robust against
// |mean| >> std as the shifted single pass of gn_stats_kernel.
// (four (image, group) pairs per workgroup, one per wave: a quarter of the workgroups to dispatch -- the launch is 2048
//  one-wave jobs of a microsecond each at batch 64, and most of its 5 us was getting them onto the chip)
constexpr int GNF_WAVES = 4;
__global__ void __launch_bounds__(64 * GNF_WAVES) gn_finalize_kernel(const float4* __restrict__ st0, int c0, int slots0,
                                                         const float4* __restrict__ st1, int c1, int slots1,
                                                         int groups, int n_jobs, float eps, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ scale,
                                                         float* __restrict__ shift, float* __restrict__ mean_rstd) {
    const int C = c0 + c1, gs = C / groups;
    const int job = blockIdx.x * GNF_WAVES + (threadIdx.x >> 6);
    if (job >= n_jobs) return;               // whole waves leave: no barrier below
    const int b = job / groups, g = job % groups;
    const int lane = threadIdx.x & 63;
    // the group's channels are one contiguous run of partials in each producer's buffer ([b][c][slot])
    const int ca = g * gs, cb = ca + gs;
    const int a0 = min(ca, c0), b0 = min(cb, c0);                 // channels [a0, b0) of the first producer
    const int a1 = max(ca, c0) - c0, b1 = max(cb, c0) - c0;       // channels [a1, b1) of the second
    const float4* run0 = st0 + ((size_t)b * c0 + a0) * slots0;
    const int len0 = (b0 - a0) * slots0;
    const float4* run1 = st1 ? st1 + ((size_t)b * c1 + a1) * slots1 : nullptr;
    const int len1 = st1 ? (b1 - a1) * slots1 : 0;
    // this lane's output channel (gs <= 64 in every network here; the tail loop below covers larger groups)
    const float my_gamma = lane < gs ? gamma[ca + lane] : 0.0f, my_beta = lane < gs ? beta[ca + lane] : 0.0f;
    constexpr int KEEP = 4;                                        // partials per lane and producer held in registers
    float4 k0[KEEP], k1[KEEP];
    double n = 0.0, s1 = 0.0, m2 = 0.0;
#pragma unroll
    for (int j = 0; j < KEEP; ++j) {
        const int i = lane + 64 * j;
        k0[j] = i < len0 ? run0[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        k1[j] = i < len1 ? run1[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < KEEP; ++j) {
        n += (double)k0[j].x + (double)k1[j].x;
        s1 += (double)k0[j].y + (double)k1[j].y;
        m2 += (double)k0[j].z + (double)k1[j].z;
    }
    for (int i = lane + 64 * KEEP; i < len0; i += 64) { const float4 v = run0[i]; n += v.x; s1 += v.y; m2 += v.z; }
    for (int i = lane + 64 * KEEP; i < len1; i += 64) { const float4 v = run1[i]; n += v.x; s1 += v.y; m2 += v.z; }
    n = wave64_sum_f64(n);
    s1 = wave64_sum_f64(s1);
    m2 = wave64_sum_f64(m2);
    const double mean = s1 / n;
    double between = 0.0;
    auto dev = [&](const float4 v) { between += gn_between_term(v.x, v.y, mean); };       // (gn_merge.h)
#pragma unroll
    for (int j = 0; j < KEEP; ++j) { dev(k0[j]); dev(k1[j]); }
    for (int i = lane + 64 * KEEP; i < len0; i += 64) dev(run0[i]);
    for (int i = lane + 64 * KEEP; i < len1; i += 64) dev(run1[i]);
    between = wave64_sum_f64(between);
    float meanf, rstd;
    gn_mean_rstd(n, s1, m2, between, eps, meanf, rstd, mean);
    if (mean_rstd && lane == 0) {
        mean_rstd[2 * (size_t)job] = meanf;
        mean_rstd[2 * (size_t)job + 1] = rstd;
    }
    if (lane < gs) {
        float sc, sh;
        gn_affine(my_gamma, my_beta, meanf, rstd, sc, sh);
        scale[(size_t)b * C + ca + lane] = sc;
        shift[(size_t)b * C + ca + lane] = sh;
    }
    for (int cc = lane + 64; cc < gs; cc += 64) {
        const int c = ca + cc;
        float sc, sh;
        gn_affine(gamma[c], beta[c], meanf, rstd, sc, sh);
        scale[(size_t)b * C + c] = sc;
        shift[(size_t)b * C + c] = sh;
    }
}

int launch_gn_finalize(sisic_ctx* ctx, const float* st0, int c0, int slots0, const float* st1, int c1, int slots1, int B,
                       int HW, int groups, float eps, const float* gamma, const float* beta, float* scale, float* shift,
                       hipStream_t s, float* mean_rstd) {
    SISIC_REQUIRE(st0 && gamma && beta && scale && shift, "groupnorm_finalize: null tensor");
    SISIC_REQUIRE((c1 == 0) == (st1 == nullptr), "groupnorm_finalize: stats1/c1 mismatch");
    const int C = c0 + c1;
    SISIC_REQUIRE(B > 0 && HW > 0 && groups > 0 && c0 > 0 && C % groups == 0 && slots0 > 0 && (c1 == 0 || slots1 > 0),
                  "groupnorm_finalize: C=%d groups=%d slots=%d/%d", C, groups, slots0, slots1);
    (void)HW;   // the element count travels with the partials
    ProfileScope prof(ctx, s, PK_GN, 16.0 * B * (c0 * (double)slots0 + c1 * (double)slots1) + 8.0 * B * C, 0.0);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(B * groups, GNF_WAVES)), dim3(64 * GNF_WAVES), 0, s,
                       reinterpret_cast<const float4*>(st0), c0, slots0, reinterpret_cast<const float4*>(st1), c1, slots1, groups,
                       B * groups, eps, gamma, beta, scale, shift, mean_rstd);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

}  // namespace sisic
