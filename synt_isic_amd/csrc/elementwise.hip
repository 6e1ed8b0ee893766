// elementwise.hip -- the HBM-bound and tiny kernels of the sampling loop:
//   * ddpm_step_kernel  : fused DDPMScheduler.step, bit-exact vs torch CPU
//   * denorm_u8_kernel  : clamp((x+1)/2,0,1)*255 -> uint8 HWC (image_generator.py:441-447)
//   * temb_mlp_kernel   : sinusoidal timestep embedding -> Linear -> SiLU -> Linear -> SiLU
//   * linear_t_kernel   : every ResnetBlock2D.time_emb_proj in one launch
//   * transpose2d_kernel: weight re-layout at load time
#include "common.h"
#include "pack_device.h"

namespace sisic {

// ---- DDPM step -------------------------------------------------------------------------
// One IEEE rounding per operation and NO FMA contraction (this file is built with
// -ffp-contract=off, and the pragma pins it locally): that is what the reference's sequence of
// separate torch ops produces, so the step is bit-exact against torch on the CPU.
__device__ __forceinline__ float ddpm_one(float e, float x, float z, float sb, float sa, float c0, float c1,
                                          float sigma, float clip, bool noise) {
#pragma clang fp contract(off)
    float x0 = (x - sb * e) / sa;
    if (clip > 0.0f) x0 = fminf(fmaxf(x0, -clip), clip);
    float r = c0 * x0 + c1 * x;
    if (noise) r = r + sigma * z;
    return r;
}

__global__ void __launch_bounds__(256)
ddpm_step_kernel(const float* __restrict__ eps, const float* x, const float* __restrict__ z,
                 float* out, int64_t n, float sb, float sa, float c0, float c1, float sigma, float clip,
                 int vec4) {
    const bool noise = (z != nullptr) && (sigma != 0.0f);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec4) {
        const int64_t n4 = n >> 2;
        const float4* e4 = reinterpret_cast<const float4*>(eps);
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* z4 = reinterpret_cast<const float4*>(z);
        float4* o4 = reinterpret_cast<float4*>(out);
        for (int64_t i = t0; i < n4; i += stride) {
            const float4 e = e4[i], xv = x4[i];
            float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (noise) zv = z4[i];
            float4 r;
            r.x = ddpm_one(e.x, xv.x, zv.x, sb, sa, c0, c1, sigma, clip, noise);
            r.y = ddpm_one(e.y, xv.y, zv.y, sb, sa, c0, c1, sigma, clip, noise);
            r.z = ddpm_one(e.z, xv.z, zv.z, sb, sa, c0, c1, sigma, clip, noise);
            r.w = ddpm_one(e.w, xv.w, zv.w, sb, sa, c0, c1, sigma, clip, noise);
            o4[i] = r;
        }
        for (int64_t i = (n4 << 2) + t0; i < n; i += stride)
            out[i] = ddpm_one(eps[i], x[i], noise ? z[i] : 0.f, sb, sa, c0, c1, sigma, clip, noise);
    } else {
        for (int64_t i = t0; i < n; i += stride)
            out[i] = ddpm_one(eps[i], x[i], noise ? z[i] : 0.f, sb, sa, c0, c1, sigma, clip, noise);
    }
}

int launch_ddpm_step(sisic_ctx* ctx, const float* eps, const float* x, const float* z, float* out, int64_t n, float sb,
                     float sa, float c0, float c1, float sigma, float clip, hipStream_t s) {
    SISIC_REQUIRE(eps && x && out && n > 0, "ddpm_step: null tensor or empty");
    SISIC_REQUIRE(sa != 0.0f, "ddpm_step: sqrt_alpha_prod is zero");
    const bool noise = z != nullptr && sigma != 0.0f;
    ProfileScope prof(ctx, s, PK_DDPM, (noise ? 16.0 : 12.0) * (double)n, 0.0);
    const uintptr_t al = reinterpret_cast<uintptr_t>(eps) | reinterpret_cast<uintptr_t>(x) |
                         reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(out);
    const int vec4 = (al & 15) == 0;
    const int64_t work = vec4 ? (n + 3) / 4 : n;
    const int blocks = (int)std::min<int64_t>((work + 255) / 256, 2048);
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(blocks), dim3(256), 0, s, eps, x, z, out, n, sb, sa, c0, c1, sigma, clip,
                       vec4);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// ---- the same step with its per-step parameters read from device memory (graph-replayed sampling loop) ----------------
// One captured step is replayed for every step of the loop, so nothing that changes from step to step may be a launch
// argument: the loop keeps {step index, noise base pointer} and its per-step tables (coefficients, noise row of the step
// or -1) in device memory; this kernel selects its row, ddpm_advance_kernel moves the index on.
struct LoopState {
    int step;
    int pad;
    const float* noise;       // base of the [n_noise, n] noise rows of this call
};

__global__ void __launch_bounds__(256)
ddpm_step_indexed_kernel(const float* __restrict__ eps, float* x, int64_t n, const LoopState* __restrict__ st,
                         const float* __restrict__ coef, const int* __restrict__ zrow, float clip, int vec4) {
    const int step = st->step;
    const float sb = coef[5 * step + 0], sa = coef[5 * step + 1], c0 = coef[5 * step + 2], c1 = coef[5 * step + 3],
                sigma = coef[5 * step + 4];
    const int zr = zrow[step];
    const float* z = zr >= 0 ? st->noise + (int64_t)zr * n : nullptr;
    const bool noise = (z != nullptr) && (sigma != 0.0f);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec4 && ((reinterpret_cast<uintptr_t>(z) & 15) == 0)) {
        const int64_t n4 = n >> 2;
        const float4* e4 = reinterpret_cast<const float4*>(eps);
        float4* x4 = reinterpret_cast<float4*>(x);
        const float4* z4 = reinterpret_cast<const float4*>(z);
        for (int64_t i = t0; i < n4; i += stride) {
            const float4 e = e4[i], xv = x4[i];
            float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (noise) zv = z4[i];
            float4 r;
            r.x = ddpm_one(e.x, xv.x, zv.x, sb, sa, c0, c1, sigma, clip, noise);
            r.y = ddpm_one(e.y, xv.y, zv.y, sb, sa, c0, c1, sigma, clip, noise);
            r.z = ddpm_one(e.z, xv.z, zv.z, sb, sa, c0, c1, sigma, clip, noise);
            r.w = ddpm_one(e.w, xv.w, zv.w, sb, sa, c0, c1, sigma, clip, noise);
            x4[i] = r;
        }
        for (int64_t i = (n4 << 2) + t0; i < n; i += stride)
            x[i] = ddpm_one(eps[i], x[i], noise ? z[i] : 0.f, sb, sa, c0, c1, sigma, clip, noise);
    } else {
        for (int64_t i = t0; i < n; i += stride)
            x[i] = ddpm_one(eps[i], x[i], noise ? z[i] : 0.f, sb, sa, c0, c1, sigma, clip, noise);
    }
}

// tproj_cur[r] = tproj_table[step][r]: the time-embedding projections of the step about to run
__global__ void loop_select_row_kernel(const float* __restrict__ table, int R, const LoopState* __restrict__ st,
                                       float* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) out[r] = table[(size_t)st->step * R + r];
}

__global__ void loop_advance_kernel(LoopState* st) { st->step += 1; }

int launch_loop_select_row(sisic_ctx*, const float* table, int R, const void* state, float* out, hipStream_t s) {
    hipLaunchKernelGGL(loop_select_row_kernel, dim3(cdiv(R, 256)), dim3(256), 0, s, table, R, static_cast<const LoopState*>(state), out);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

int launch_loop_advance(sisic_ctx*, void* state, hipStream_t s) {
    hipLaunchKernelGGL(loop_advance_kernel, dim3(1), dim3(1), 0, s, static_cast<LoopState*>(state));
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

int launch_ddpm_step_indexed(sisic_ctx* ctx, const float* eps, float* x, int64_t n, const void* state, const float* coef,
                             const int* zrow, float clip, hipStream_t s) {
    SISIC_REQUIRE(eps && x && state && coef && zrow && n > 0, "ddpm_step_indexed: null argument");
    ProfileScope prof(ctx, s, PK_DDPM, 16.0 * (double)n, 0.0);
    const uintptr_t al = reinterpret_cast<uintptr_t>(eps) | reinterpret_cast<uintptr_t>(x);
    const int vec4 = (al & 15) == 0 && (n & 3) == 0;
    const int64_t work = vec4 ? (n + 3) / 4 : n;
    const int blocks = (int)std::min<int64_t>((work + 255) / 256, 2048);
    hipLaunchKernelGGL(ddpm_step_indexed_kernel, dim3(blocks), dim3(256), 0, s, eps, x, n, static_cast<const LoopState*>(state), coef,
                       zrow, clip, vec4);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

size_t loop_state_bytes() { return sizeof(LoopState); }

// ---- de-normalise to uint8 HWC ---------------------------------------------------------------
// FORM 0: image_generator.py:441-447     clamp((x + 1) / 2, 0, 1) * 255, truncated
// FORM 1: generate_test.py:94-97          (clamp(x, -1, 1) + 1) * 0.5 * 255, truncated   (bit-equal to form 0)
// FORM 2: diffusion_generator.py:231-232  clip((x + 1) * 127.5, 0, 255), truncated      (rounds differently: one
//         multiplication by 127.5 instead of a halving and a multiplication by 255)
// fp32 operation order of the respective torch / numpy expressions; this file is compiled without FMA contraction.
template <int FORM>
__global__ void __launch_bounds__(256)
denorm_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int C, int HW, int64_t total) {
    // one thread per output byte: index = (b*HW + p)*C + c
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t bp = i / C;
        const int p = (int)(bp % HW);
        const int64_t b = bp / HW;
        float v = x[(b * C + c) * HW + p];
        if constexpr (FORM == 0) {
            v = (v + 1.0f) / 2.0f;
            v = fminf(fmaxf(v, 0.0f), 1.0f);
            v = v * 255.0f;
        } else if constexpr (FORM == 1) {
            v = fminf(fmaxf(v, -1.0f), 1.0f);
            v = (v + 1.0f) * 0.5f;
            v = v * 255.0f;
        } else {
            v = (v + 1.0f) * 127.5f;
            v = fminf(fmaxf(v, 0.0f), 255.0f);
        }
        out[i] = (uint8_t)(int)v;
    }
}

int launch_denorm_u8(sisic_ctx* ctx, const float* x, uint8_t* out, int B, int C, int H, int W, hipStream_t s, int form) {
    SISIC_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0, "denorm_u8: bad arguments");
    SISIC_REQUIRE(form >= 0 && form <= 2, "denorm_u8: form %d (0 = image_generator, 1 = generate_test, 2 = diffusion_generator)", form);
    const int64_t total = (int64_t)B * C * H * W;
    ProfileScope prof(ctx, s, PK_OTHER, 5.0 * (double)total, 0.0);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);
    if (form == 0) hipLaunchKernelGGL(denorm_u8_kernel<0>, dim3(blocks), dim3(256), 0, s, x, out, C, H * W, total);
    else if (form == 1) hipLaunchKernelGGL(denorm_u8_kernel<1>, dim3(blocks), dim3(256), 0, s, x, out, C, H * W, total);
    else hipLaunchKernelGGL(denorm_u8_kernel<2>, dim3(blocks), dim3(256), 0, s, x, out, C, H * W, total);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// ---- time embedding -----------------------------------------------------------------------------
__device__ __forceinline__ float silu_acc(float v) { return v / (1.0f + expf(-v)); }

// one workgroup per sample; hidden <= 1024; weights stored transposed [in][hidden]
__global__ void __launch_bounds__(256)
temb_mlp_kernel(const float* __restrict__ t_vals, const float* __restrict__ freqs, int n_freqs,
                const float* __restrict__ w1t, const float* __restrict__ b1, const float* __restrict__ w2t,
                const float* __restrict__ b2, int hidden, float* __restrict__ temb_act, float* __restrict__ save_emb,
                float* __restrict__ save_h1, float* __restrict__ save_t2) {
    __shared__ float e[256];
    __shared__ float h[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float t = t_vals[b];
    const int nin = 2 * n_freqs;
    for (int k = tid; k < nin; k += blockDim.x) {
        const float arg = t * freqs[k % n_freqs];
        e[k] = (k < n_freqs) ? cosf(arg) : sinf(arg);   // flip_sin_to_cos=True: cos half first
        if (save_emb) save_emb[(size_t)b * nin + k] = e[k];
    }
    __syncthreads();
    for (int j = tid; j < hidden; j += blockDim.x) {
        float acc = 0.0f;
        for (int k = 0; k < nin; ++k) acc += w1t[(size_t)k * hidden + j] * e[k];
        h[j] = silu_acc(acc + b1[j]);
        if (save_h1) save_h1[(size_t)b * hidden + j] = acc + b1[j];
    }
    __syncthreads();
    for (int j = tid; j < hidden; j += blockDim.x) {
        float acc = 0.0f;
        for (int k = 0; k < hidden; ++k) acc += w2t[(size_t)k * hidden + j] * h[k];
        temb_act[(size_t)b * hidden + j] = silu_acc(acc + b2[j]);
        if (save_t2) save_t2[(size_t)b * hidden + j] = acc + b2[j];
    }
}

int launch_temb_mlp(sisic_ctx* ctx, const float* t_vals, int B, const float* freqs, int n_freqs, const float* w1t,
                    const float* b1, const float* w2t, const float* b2, int hidden, float* temb_act, hipStream_t s,
                    float* save_emb, float* save_h1, float* save_t2) {
    SISIC_REQUIRE(n_freqs > 0 && 2 * n_freqs <= 256 && hidden > 0 && hidden <= 1024, "temb_mlp: sizes unsupported");
    ProfileScope prof(ctx, s, PK_OTHER, 0.0, 0.0);
    hipLaunchKernelGGL(temb_mlp_kernel, dim3(B), dim3(256), 0, s, t_vals, freqs, n_freqs, w1t, b1, w2t, b2, hidden,
                       temb_act, save_emb, save_h1, save_t2);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// out[b, r] = bias[r] + sum_k wt[k][r] * x[b][k]
__global__ void __launch_bounds__(256)
linear_t_kernel(const float* __restrict__ x, int K, const float* __restrict__ wt, const float* __restrict__ bias,
                int R, float* __restrict__ out) {
    __shared__ float xs[1024];
    const int b = blockIdx.y;
    for (int k = threadIdx.x; k < K; k += blockDim.x) xs[k] = x[(size_t)b * K + k];
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) {
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc += wt[(size_t)k * R + r] * xs[k];
        out[(size_t)b * R + r] = acc + (bias ? bias[r] : 0.0f);
    }
}

int launch_linear_t(sisic_ctx* ctx, const float* x, int B, int K, const float* wt, const float* bias, int R,
                    float* out, hipStream_t s) {
    SISIC_REQUIRE(K > 0 && K <= 1024 && R > 0 && B > 0, "linear_t: sizes unsupported");
    ProfileScope prof(ctx, s, PK_OTHER, 0.0, 0.0);
    hipLaunchKernelGGL(linear_t_kernel, dim3(cdiv(R, 256), B), dim3(256), 0, s, x, K, wt, bias, R, out);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// out[c * out_ld + out_col0 + r] = in[r * cols + c]
__global__ void transpose2d_kernel(const float* __restrict__ in, int rows, int cols, float* __restrict__ out,
                                   int out_ld, int out_col0) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        transpose2d_elem((size_t)i, in, cols, out, out_ld, out_col0);
    }
}

int launch_transpose2d(sisic_ctx*, const float* in, int rows, int cols, float* out, int out_ld, int out_col0,
                       hipStream_t s) {
    const int64_t total = (int64_t)rows * cols;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 1024);
    hipLaunchKernelGGL(transpose2d_kernel, dim3(blocks), dim3(256), 0, s, in, rows, cols, out, out_ld, out_col0);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

}  // namespace sisic
