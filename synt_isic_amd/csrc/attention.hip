// attention.hip -- multi-head self-attention core for 32 heads x d=8 (NCHW token layout).
//
// Replaces scaled_dot_product_attention inside diffusers' Attention block as configured by
// the reference (attention_head_dim=8).  The q/k/v and output
// projections are 1x1 convolutions (conv_mfma.hip) that leave q,k,v as [B, 3C, N] with the
// token index contiguous, so every operand here is read as contiguous rows.
//
// One wave owns 32 or 64 queries of one (sample, head) (attention_kernel<QB>); a workgroup (4 waves) owns 128 or 256
// queries and shares the head's K/V rows through LDS in blocks of 256 keys.
//   * S^T = K^T Q with fp32-EQUIVALENT products on the bf16 matrix pipe (round 3; the arithmetic of conv_winograd_bf3.inc:
//     q and k split exactly into three bf16 terms, six of the nine term products in three v_mfma_f32_32x32x16_bf16 --
//     K = 16 is the eight head dimensions x two terms -- 96 matrix cycles per 32 x 32 score tile instead of the 256 of four
//     v_mfma_f32_32x32x2_f32, and on a pipe of its own: the f32 MFMA IS the vector unit this kernel is bound by).  K is
//     split once per key block while it is staged into LDS, q once per wave.  Computed "swapped" so the query sits on
//     the lane and its keys in the 16 accumulator registers: the softmax row reduction is register-local plus one
//     cross-half shuffle;
//   * softmax in fp32, online (running max / sum / output) across 32-key tiles and key blocks;
//   * P.V on the vector ALU: with d=8 the MFMA tile would be 3/4 padding (and fp32 MFMA is only twice the packed
//     vector rate).  V sits in LDS as [key][d], so one key's eight values are two broadcast ds_read_b128 whose
//     register pairs feed v_pk_fma_f32 directly: four packed FMAs per (query, key), no operand shuffling.
//
// Algorithmic bytes per launch: 4*B*4*C*N (q,k,v in, o out).  FLOPs: 4*B*C*N*N.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace sisic {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short att_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned att_u4 __attribute__((ext_vector_type(4)));
typedef unsigned att_u2 __attribute__((ext_vector_type(2)));

// x = hi + mid + lo exactly (8 + 8 + 8 significant bits); the terms of two values packed [b : a] per dword
__device__ __forceinline__ void att_split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned ha = __float_as_uint(a) & 0xffff0000u, hb = __float_as_uint(b) & 0xffff0000u;
    const float ra = a - __uint_as_float(ha), rb = b - __uint_as_float(hb);
    const unsigned ma = __float_as_uint(ra) & 0xffff0000u, mb = __float_as_uint(rb) & 0xffff0000u;
    const float la = ra - __uint_as_float(ma), lb = rb - __uint_as_float(mb);
    hi = __builtin_amdgcn_perm(hb, ha, 0x07060302u);
    mid = __builtin_amdgcn_perm(mb, ma, 0x07060302u);
    lo = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

constexpr int ATT_D = 8;
constexpr int ATT_KB = 256;       // keys per LDS block
constexpr int ATT_KT = ATT_KB / 32;
constexpr int ATT_WAVES = 4;

// QB: 32-query blocks per wave.  With two, a key's K operands and its eight V values -- two broadcast ds_read_b128, the LDS
// traffic this loop is bound by -- are read once for 64 queries, and a workgroup stages the head's K / V once for 256 queries.
// A query's arithmetic does not depend on QB (same keys, same order, same operations): the choice may follow the batch.
template <int QB>
__global__ void __launch_bounds__(64 * ATT_WAVES)
attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, int N, int heads, int q_blocks,
                 float scale) {
    // K of the block, split: per dimension group g (d = 4 g .. 4 g + 3) and key the packed terms (hi, mid) [4 dwords] and lo [2]
    __shared__ __attribute__((aligned(16))) unsigned Kh[2 * ATT_KB * 4];
    __shared__ __attribute__((aligned(16))) unsigned Kl[2 * ATT_KB * 2];
    __shared__ __attribute__((aligned(16))) float Vs[ATT_D * ATT_KB];

    int blk = blockIdx.x;
    const int qb = blk % q_blocks;
    blk /= q_blocks;
    const int head = blk % heads;
    const int b = blk / heads;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int q0 = qb * (32 * QB * ATT_WAVES) + wave * (32 * QB);     // the wave's queries: q0 + 32 j + l31
    const bool wave_active = q0 < N;      // wave-uniform

    const float* Qp = qkv + ((size_t)b * 3 * C + (size_t)head * ATT_D) * N;
    const float* Kp = Qp + (size_t)C * N;
    const float* Vp = Kp + (size_t)C * N;

    // the softmax scale AND log2(e) are folded into q once: the score tile needs no multiply and p = 2^(s - m) is a bare
    // v_exp_f32 (softmax is invariant under the common base change)
    // lane = (query, dimension group g = half): the B operands of the three MFMAs, (q_hi, q_mid), (q_mid, q_hi), (q_lo, q_hi)
    att_u4 b_hm[QB], b_mh[QB], b_lh[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const int q = q0 + 32 * j + l31;
        float qv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) qv[k] = (q < N) ? Qp[(size_t)(4 * half + k) * N + q] * scale : 0.0f;
        unsigned h0, m0, l0, h1, m1, l1;
        att_split2(qv[0], qv[1], h0, m0, l0);
        att_split2(qv[2], qv[3], h1, m1, l1);
        b_hm[j] = att_u4{h0, h1, m0, m1};
        b_mh[j] = att_u4{m0, m1, h0, h1};
        b_lh[j] = att_u4{l0, l1, h0, h1};
    }

    float m_run[QB], l_part[QB];
    f32x2 o2[QB][ATT_D / 2];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        m_run[j] = -INFINITY; l_part[j] = 0.0f;
#pragma unroll
        for (int d = 0; d < ATT_D / 2; ++d) o2[j][d] = f32x2{0.0f, 0.0f};
    }

    // One pass per 32-key tile with an online softmax (running maximum m_run, running sum l_part, running
    // output o2[]): S^T tile on the MFMA pipe, tile maximum, rescale of the running state when the maximum
    // grows, p = exp(s - m), row sum and P.V.  (A two-pass form that recomputed S^T spent twice the matrix
    // work for the same result; on this chip matrix and vector instructions of one SIMD do not overlap.)
    // MASKED: the tile reaches past the last key (only the final tile of a sequence that is no multiple of 32).
    auto tile = [&](const int kt, const int nk, auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
        f32x16 S[QB];
        {   // k_hi q_hi + k_mid q_mid, k_hi q_mid + k_mid q_hi, k_hi q_lo + k_lo q_hi
            const int key = half * ATT_KB + kt * 32 + l31;
            const att_u4 a_hm = *reinterpret_cast<const att_u4*>(&Kh[key * 4]);
            const att_u2 a_l = *reinterpret_cast<const att_u2*>(&Kl[key * 2]);
            const att_u4 a_hl = {a_hm.x, a_hm.y, a_l.x, a_l.y};
#pragma unroll
            for (int j = 0; j < QB; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[j][r] = 0.0f;
                S[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(att_bf16x8, a_hm), __builtin_bit_cast(att_bf16x8, b_hm[j]), S[j], 0, 0, 0);
                S[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(att_bf16x8, a_hm), __builtin_bit_cast(att_bf16x8, b_mh[j]), S[j], 0, 0, 0);
                S[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(att_bf16x8, a_hl), __builtin_bit_cast(att_bf16x8, b_lh[j]), S[j], 0, 0, 0);
            }
        }
        float m_new[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (MASKED) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    S[j][r] = (key < nk) ? S[j][r] : -INFINITY;
                }
                tmax = fmaxf(tmax, S[j][r]);
            }
            {   // the other half's maximum: v_permlane32_swap (one instruction; __shfl_xor is an LDS round trip)
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
                tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            m_new[j] = fmaxf(m_run[j], tmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new[j]);     // first tile: 2^(-inf) = 0; unchanged maximum: 1
            // (multiplications the compiler cannot fuse with the additions that follow: fused -- l * alpha + sum, o * alpha + p v --
            //  by one instantiation and not by the other, the two forms' bits would differ)
            // (s_nop: alpha comes from v_exp_f32, and the hazard recognizer does not look inside an asm -- a vector instruction
            //  that reads a transcendental's result in the next slot reads the old register; found as garbage in the masked tile)
            asm("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(l_part[j]) : "v"(l_part[j]), "v"(alpha));
            const f32x2 a2 = f32x2{alpha, alpha};
#pragma unroll
            for (int d = 0; d < ATT_D / 2; ++d) asm("s_nop 1\n\tv_pk_mul_f32 %0, %1, %2" : "=v"(o2[j][d]) : "v"(o2[j][d]), "v"(a2));
            m_run[j] = m_new[j];
        }
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float pv[QB][4];
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                // (two subtractions per instruction: v_pk_add_f32 with the maximum negated -- this loop is bound by its vector instructions)
                const f32x2 mm = f32x2{m_new[j], m_new[j]};
                const f32x2 d0 = f32x2{S[j][4 * rq], S[j][4 * rq + 1]} - mm, d1 = f32x2{S[j][4 * rq + 2], S[j][4 * rq + 3]} - mm;
                pv[j][0] = __builtin_amdgcn_exp2f(d0.x); pv[j][1] = __builtin_amdgcn_exp2f(d0.y);           // masked keys: 2^(-inf) = 0
                pv[j][2] = __builtin_amdgcn_exp2f(d1.x); pv[j][3] = __builtin_amdgcn_exp2f(d1.y);
                l_part[j] += (pv[j][0] + pv[j][1]) + (pv[j][2] + pv[j][3]);
            }
            const int koff = kt * 32 + 8 * rq + 4 * half;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 va = *reinterpret_cast<const float4*>(&Vs[(koff + i) * ATT_D]);
                const float4 vb = *reinterpret_cast<const float4*>(&Vs[(koff + i) * ATT_D + 4]);
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    const f32x2 p2 = f32x2{pv[j][i], pv[j][i]};
                    o2[j][0] += p2 * f32x2{va.x, va.y};
                    o2[j][1] += p2 * f32x2{va.z, va.w};
                    o2[j][2] += p2 * f32x2{vb.x, vb.y};
                    o2[j][3] += p2 * f32x2{vb.z, vb.w};
                }
            }
        }
    };

    for (int kb0 = 0; kb0 < N; kb0 += ATT_KB) {
        __syncthreads();
        for (int idx = tid; idx < 2 * ATT_KB; idx += 64 * ATT_WAVES) {         // K: (dimension group, key) -> its split terms
            const int g = idx / ATT_KB, k = idx % ATT_KB;
            const int key = kb0 + k;
            float kv[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) kv[d] = key < N ? Kp[(size_t)(4 * g + d) * N + key] : 0.0f;
            unsigned h0, m0, l0, h1, m1, l1;
            att_split2(kv[0], kv[1], h0, m0, l0);
            att_split2(kv[2], kv[3], h1, m1, l1);
            *reinterpret_cast<att_u4*>(&Kh[idx * 4]) = att_u4{h0, h1, m0, m1};
            *reinterpret_cast<att_u2*>(&Kl[idx * 2]) = att_u2{l0, l1};
        }
        for (int k = tid; k < ATT_KB; k += 64 * ATT_WAVES) {                   // V: [key][d], one key per thread
            const int key = kb0 + k;
            float v[ATT_D];
#pragma unroll
            for (int d = 0; d < ATT_D; ++d) v[d] = key < N ? Vp[(size_t)d * N + key] : 0.0f;
            *reinterpret_cast<float4*>(&Vs[k * ATT_D]) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(&Vs[k * ATT_D + 4]) = make_float4(v[4], v[5], v[6], v[7]);
        }
        __syncthreads();
        if (!wave_active) continue;
        const int nk = min(ATT_KB, N - kb0);
#pragma unroll 1
        for (int kt = 0; kt * 32 < nk; ++kt) {
            if (kt * 32 + 32 <= nk) tile(kt, nk, std::false_type{});
            else tile(kt, nk, std::true_type{});
        }
    }

    if (wave_active) {
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            const int q = q0 + 32 * j + l31;
            float o[ATT_D];
#pragma unroll
            for (int d = 0; d < ATT_D / 2; ++d) { o[2 * d] = o2[j][d].x; o[2 * d + 1] = o2[j][d].y; }
            const float l_tot = l_part[j] + __shfl_xor(l_part[j], 32, 64);
            const float inv = 1.0f / l_tot;
            float res[ATT_D];
#pragma unroll
            for (int d = 0; d < ATT_D; ++d) res[d] = (o[d] + __shfl_xor(o[d], 32, 64)) * inv;
            if (q < N) {
                float* Op = out + ((size_t)b * C + (size_t)head * ATT_D) * N + q;
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    // bitwise select: a plain ?: on the array makes hipcc index it through scratch memory
                    const int hm = -half;
                    const float v = __int_as_float((__float_as_int(res[dd]) & ~hm) | (__float_as_int(res[4 + dd]) & hm));
                    Op[(size_t)(4 * half + dd) * N] = v;
                }
            }
        }
    }
}

int launch_attention(sisic_ctx* ctx, const float* qkv, float* out, int B, int C, int N, int head_dim, hipStream_t s) {
    SISIC_REQUIRE(qkv && out, "attention: null tensor");
    SISIC_REQUIRE(head_dim == ATT_D, "attention: head_dim %d unsupported (the reference uses 8)", head_dim);
    SISIC_REQUIRE(B > 0 && N > 0 && C > 0 && C % head_dim == 0, "attention: bad shape B=%d C=%d N=%d", B, C, N);
    const int heads = C / head_dim;
    // two query blocks per wave where that still leaves every CU four workgroups (SISIC_ATT_QB=1|2 forces one form: same bits)
    static const int qb_env = [] { const char* e = std::getenv("SISIC_ATT_QB"); return e ? std::atoi(e) : 0; }();
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    const int qb = qb_env == 1 || qb_env == 2 ? qb_env : ((int64_t)B * heads * cdiv(N, 64 * ATT_WAVES) >= 4 * (int64_t)cus ? 2 : 1);
    const int q_blocks = cdiv(N, 32 * qb * ATT_WAVES);
    const int64_t grid = (int64_t)B * heads * q_blocks;
    SISIC_REQUIRE(grid < (int64_t(1) << 31), "attention: grid too large");
    ProfileScope prof(ctx, s, PK_ATTN, 16.0 * B * C * N, 4.0 * B * C * double(N) * N);
    const float scale = 1.4426950408889634f / sqrtf((float)head_dim);     // head_dim^-1/2 * log2(e)
    if (qb == 2)
        hipLaunchKernelGGL(attention_kernel<2>, dim3((unsigned)grid), dim3(64 * ATT_WAVES), 0, s, qkv, out, C, N, heads, q_blocks, scale);
    else
        hipLaunchKernelGGL(attention_kernel<1>, dim3((unsigned)grid), dim3(64 * ATT_WAVES), 0, s, qkv, out, C, N, heads, q_blocks, scale);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

}  // namespace sisic
