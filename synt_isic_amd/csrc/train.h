// train.h -- launchers of train_kernels.hip (used by train.cpp and by the single-operator C entry points).
#pragma once

#include "common.h"

namespace sisic {

// dW[co][ci][ky][kx] = sum_{b,oy,ox} dy[b,co,oy,ox] * act(cat(in0,in1))[b,ci,oy*stride+ky-pad, ox*stride+kx-pad]
// with the forward convolution's prologue act (GroupNorm apply + optional SiLU, zero padding after it) and index maps.
struct WgradArgs {
    const float* in0 = nullptr; const float* in1 = nullptr;
    int c0 = 0, c1 = 0, B = 0, Hin = 0, Win = 0;
    int ups = 0;                 // nearest 2x before the convolution
    int ksize = 3, stride = 1;
    const float* gn_scale = nullptr; const float* gn_shift = nullptr; int gn_silu = 0;
    const float* dy = nullptr;   // [B, Cout, Hout, Wout]
    int Cout = 0;
    float* dw = nullptr;         // [Cout, c0+c1, k, k] (OIHW, the state-dict layout)
    float* dw1 = nullptr;        // 1x1 only, with dw2: the gradient leaves as three [Cout/3, Cin] tensors (fused q/k/v projection)
    float* dw2 = nullptr;
};
size_t conv_wgrad_scratch_floats(const WgradArgs& a);
int launch_conv_wgrad(sisic_ctx*, const WgradArgs& a, float* part, size_t part_floats, hipStream_t s);
int launch_transpose_flip(sisic_ctx*, const float* w, int Cout, int Cin, int KK, float* wt, hipStream_t s);

// repack.hip: one re-layout of one tensor inside a batched launch (pack_batch_kernel)
struct PackJob {
    enum Kind : int { COPY = 0, TRANSPOSE2D, FLIP, CONV_PACK, WINO_FIRST, WINO_WIDE, WINO_BF3 };
    int kind;
    int a, b, c, d, e;          // shape arguments of the kind's per-element function (see the constructors in repack.hip)
    const float* src;
    float* dst;
    size_t total;               // elements of the job's index space
    int first_block;            // first workgroup of the launch that works on this job
};
int pack_job_blocks(const PackJob& j);
int launch_pack_batch(sisic_ctx*, const PackJob* dev_jobs, int njobs, int nblocks, hipStream_t s);
PackJob pack_job_copy(const float* src, float* dst, size_t n);
PackJob pack_job_transpose2d(const float* in, int rows, int cols, float* out, int out_ld, int out_col0);
PackJob pack_job_flip(const float* w, int Cout, int Cin, int KK, float* wt);
PackJob pack_job_conv(const float* w, int Cout, int Cin, int k, float* packed);
PackJob pack_job_wino_first(const float* w, int Cout, int Cin, float* packed);
PackJob pack_job_wino_wide(int Cout, int Cin, float* packed);
PackJob pack_job_wino_bf3(int Cout, int Cin, float* packed);
int launch_plane_sums(sisic_ctx*, const float* x, int planes, int HW, float* out, hipStream_t s);
// bias gradient of a convolution in one launch: db[c] = sum over (b, pixels) of dy; split > 0: channels [0, split) -> db0,
// [split, 2 split) -> db1, the rest -> db2; tproj (optional): the per-(image, channel) sums into column c of [B, tproj_ld]
int launch_bias_grad(sisic_ctx*, const float* dy, int B, int C, int HW, float* db0, float* db1, float* db2, int split, float* tproj,
                     int tproj_ld, hipStream_t s);
int launch_col_sums(sisic_ctx*, const float* m, int rows, int cols, int ld, float* out, int accumulate, hipStream_t s);
int launch_copy_cols(sisic_ctx*, const float* src, int rows, int cols, float* dst, int ld_dst, hipStream_t s);
// sums: scratch [2][B][c0+c1]
int launch_gn_bwd(sisic_ctx*, const float* da, const float* in0, int c0, const float* in1, int c1, int B, int HW, int groups,
                  const float* scale, const float* shift, const float* mean_rstd, const float* gamma, int silu,
                  float* sums, float* g0, float* g1, float* dgamma, float* dbeta, hipStream_t s);
int launch_accum_split(sisic_ctx*, const float* da, int B, int C, int HW, float* g0, int c0, float* g1, int c1, hipStream_t s);
int launch_accum_pool2(sisic_ctx*, const float* da, int planes, int H, int W, float* g, hipStream_t s);
int launch_add_inplace(sisic_ctx*, float* dst, const float* src, size_t n, hipStream_t s);
int launch_attention_bwd(sisic_ctx*, const float* qkv, const float* o, const float* dO, float* dqkv, int B, int C, int N,
                         int head_dim, hipStream_t s);
int launch_linear_wgrad(sisic_ctx*, const float* dy, int ld, const float* x, int B, int R, int K, float* dW, hipStream_t s);
int launch_linear_dgrad(sisic_ctx*, const float* dy, int ld, const float* W, int B, int R, int K, float* dx, hipStream_t s,
                        int w_is_transposed);
int launch_silu_fwd(sisic_ctx*, const float* pre, size_t n, float* out, hipStream_t s);
int launch_silu_bwd(sisic_ctx*, const float* dy, const float* pre, size_t n, float* out, hipStream_t s);
int launch_mse(sisic_ctx*, const float* pred, const float* target, size_t n, float grad_scale, float* loss_dev, float* dpred,
               float* part, int nparts, hipStream_t s);
int launch_check_finite(sisic_ctx*, const float* g, size_t n, int* flag, hipStream_t s);
int launch_adam(sisic_ctx*, float* p, const float* g, float* m, float* v, size_t n, double lr, double b1, double b2, double eps,
                int64_t step, float inv_scale, hipStream_t s);
int launch_add_noise(sisic_ctx*, const float* x0, const float* noise, const float* a_dev, const float* c_dev, float* out, int B,
                     size_t per, hipStream_t s);

}  // namespace sisic
