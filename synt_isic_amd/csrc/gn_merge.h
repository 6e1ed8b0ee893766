// gn_merge.h -- the float64 arithmetic that turns GroupNorm partials (count, sum, M2 about the partial's own mean) into a group's
// (mean, rstd) and a channel's (scale, shift): ONE definition for gn_finalize_kernel (groupnorm.hip) and for the producers that
// finalize their own output (conv_winograd.hip: the 8x8 level's reduction kernel), so that both compile to the same operations
// and give the same bits.
#pragma once

namespace sisic {

// n_i (mean_i - mean)^2 = (s1_i - n_i mean)^2 / n_i of one partial; n_i is a small integer, so its fp32 reciprocal (1 ulp) only
// perturbs this term by 1e-7 relative -- no float64 division per partial.  0 for an empty partial.
__device__ __forceinline__ double gn_between_term(float cnt, float s1, double mean) {
    if (!(cnt > 0.0f)) return 0.0;
    const double d = (double)s1 - (double)cnt * mean;
    return d * d * (double)(1.0f / cnt);
}

__device__ __forceinline__ void gn_mean_rstd(double n, double s1, double m2, double between, float eps, float& meanf, float& rstd,
                                             double mean) {
    const double var = fmax((m2 + between) / n, 0.0);
    rstd = 1.0f / sqrtf((float)var + eps);
    meanf = (float)mean;
}

__device__ __forceinline__ void gn_affine(float gamma, float beta, float meanf, float rstd, float& scale, float& shift) {
    const float sc = gamma * rstd;
    scale = sc;
    shift = beta - meanf * sc;
}

}  // namespace sisic
