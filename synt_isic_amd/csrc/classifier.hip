// classifier.hip -- the small kernels around the ResNet18 classifier of xai/XAI.py:357-471:
//   * preprocess_kernel : clamp((x+1)/2,0,1) -> bilinear resize to 224x224 (align_corners=False; the
//                         reference's antialias=True is a no-op when upscaling) -> ImageNet normalise
//                         (XAI.py:399-431), one fused pass
//   * maxpool3x3s2_kernel, avgpool_fc_kernel (global average pool + Linear 512->classes)
//   * class_scores_kernel: probs[:,c] and log(probs[:,c] + 1e-8)   (XAI.py:443-471)
//   * mask_patches_kernel: the masked copies of compute_shap_approximation (XAI.py:1147-1161)
// The convolutions (7x7 s2, 3x3, 1x1 s2, BatchNorm folded, ReLU/residual epilogues) run in conv_mfma.hip.
#include "common.h"

namespace sisic {

__global__ void __launch_bounds__(256)
preprocess_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W, int OH, int OW,
                  float sh, float sw, float m0, float m1, float m2, float is0, float is1, float is2, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % OW);
        int64_t r = i / OW;
        const int oy = (int)(r % OH);
        r /= OH;
        const int c = (int)(r % C);
        const int64_t b = r / C;
        const float* src = x + (b * C + c) * (int64_t)H * W;
        float v;
        if (OH == H && OW == W) {
            v = src[(int64_t)oy * W + ox];
            v = fminf(fmaxf((v + 1.0f) / 2.0f, 0.0f), 1.0f);
        } else {
            // area_pixel_compute_source_index(scale, dst, align_corners=false): max(0, scale*(dst+0.5)-0.5)
            const float fy = fmaxf(sh * ((float)oy + 0.5f) - 0.5f, 0.0f);
            const float fx = fmaxf(sw * ((float)ox + 0.5f) - 0.5f, 0.0f);
            const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
            const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            auto px = [&](int yy, int xx) {
                const float t = src[(int64_t)yy * W + xx];
                return fminf(fmaxf((t + 1.0f) / 2.0f, 0.0f), 1.0f);      // clamp BEFORE the resize, as the reference
            };
            const float top = px(y0, x0) * (1.0f - lx) + px(y0, x1) * lx;
            const float bot = px(y1, x0) * (1.0f - lx) + px(y1, x1) * lx;
            v = top * (1.0f - ly) + bot * ly;
        }
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
        const float istd = c == 0 ? is0 : (c == 1 ? is1 : is2);
        out[i] = (v - mean) * istd;
    }
}

int launch_preprocess(sisic_ctx* ctx, const float* x, float* out, int B, int H, int W, int OH, int OW, hipStream_t s) {
    SISIC_REQUIRE(x && out && B > 0 && H > 0 && W > 0, "classifier preprocess: bad arguments");
    SISIC_REQUIRE((H == OH && W == OW) || (H <= OH && W <= OW),
                  "classifier preprocess: %dx%d -> %dx%d is a down-scale; only up-scaling (antialias no-op) is supported", H,
                  W, OH, OW);
    const int64_t total = (int64_t)B * 3 * OH * OW;
    ProfileScope prof(ctx, s, PK_OTHER, 4.0 * B * 3 * ((double)H * W + (double)OH * OW), 0.0);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(preprocess_kernel, dim3(blocks), dim3(256), 0, s, x, out, 3, H, W, OH, OW, (float)H / (float)OH,
                       (float)W / (float)OW, 0.485f, 0.456f, 0.406f, 1.0f / 0.229f, 1.0f / 0.224f, 1.0f / 0.225f, total);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// MaxPool2d(kernel 3, stride 2, padding 1)
__global__ void __launch_bounds__(256)
maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int OH, int OW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % OW);
        int64_t r = i / OW;
        const int oy = (int)(r % OH);
        const int64_t plane = r / OH;
        const float* src = x + plane * (int64_t)H * W;
        float m = -INFINITY;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int y = 2 * oy - 1 + ky;
            if (y < 0 || y >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = 2 * ox - 1 + kx;
                if (xx < 0 || xx >= W) continue;
                m = fmaxf(m, src[(int64_t)y * W + xx]);
            }
        }
        out[i] = m;
    }
}

int launch_maxpool(sisic_ctx* ctx, const float* x, float* out, int B, int C, int H, int W, hipStream_t s) {
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)B * C * OH * OW;
    ProfileScope prof(ctx, s, PK_OTHER, 4.0 * B * C * ((double)H * W + (double)OH * OW), 0.0);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(blocks), dim3(256), 0, s, x, out, H, W, OH, OW, total);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// AdaptiveAvgPool2d(1) + Linear(C -> n_out): one workgroup per sample
__global__ void __launch_bounds__(256)
avgpool_fc_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                  float* __restrict__ out, int C, int HW, int n_out) {
    __shared__ float pooled[1024];
    const int b = blockIdx.x;
    const float inv = 1.0f / (float)HW;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float* src = x + ((int64_t)b * C + c) * HW;
        float acc = 0.0f;
        for (int i = 0; i < HW; ++i) acc += src[i];
        pooled[c] = acc * inv;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = wave; o < n_out; o += blockDim.x / 64) {
        float acc = 0.0f;
        for (int c = lane; c < C; c += 64) acc += pooled[c] * w[(int64_t)o * C + c];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if (lane == 0) out[(int64_t)b * n_out + o] = acc + bias[o];
    }
}

int launch_avgpool_fc(sisic_ctx* ctx, const float* x, const float* w, const float* bias, float* out, int B, int C,
                      int HW, int n_out, hipStream_t s) {
    SISIC_REQUIRE(C <= 1024, "avgpool_fc: %d channels unsupported", C);
    ProfileScope prof(ctx, s, PK_OTHER, 4.0 * B * C * HW, 0.0);
    hipLaunchKernelGGL(avgpool_fc_kernel, dim3(B), dim3(256), 0, s, x, w, bias, out, C, HW, n_out);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// probs = softmax(logits, 1)[:, target];  logscore = log(probs + 1e-8)
__global__ void class_scores_kernel(const float* __restrict__ logits, int B, int n, int target, float* __restrict__ prob,
                                    float* __restrict__ logscore) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* l = logits + (int64_t)b * n;
    float m = l[0];
    for (int i = 1; i < n; ++i) m = fmaxf(m, l[i]);
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) sum += expf(l[i] - m);
    const float p = expf(l[target] - m) / sum;
    if (prob) prob[b] = p;
    if (logscore) logscore[b] = logf(p + 1e-8f);
}

int launch_class_scores(sisic_ctx* ctx, const float* logits, int B, int n, int target, float* prob, float* logscore,
                        hipStream_t s) {
    SISIC_REQUIRE(logits && B > 0 && n > 0 && target >= 0 && target < n, "class_scores: bad arguments");
    hipLaunchKernelGGL(class_scores_kernel, dim3(cdiv(B, 256)), dim3(256), 0, s, logits, B, n, target, prob, logscore);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// out[s, c, y, x] = mask[s, y/patch, x/patch] ? image[c, y, x] : 0        (image is one sample, masks uint8)
__global__ void __launch_bounds__(256)
mask_patches_kernel(const float* __restrict__ image, const uint8_t* __restrict__ masks, float* __restrict__ out, int C,
                    int H, int W, int patch, int ph, int pw, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        int64_t r = i / W;
        const int y = (int)(r % H);
        r /= H;
        const int c = (int)(r % C);
        const int64_t s = r / C;
        const int py = y / patch, px = x / patch;
        const bool keep = py < ph && px < pw && masks[(s * ph + py) * pw + px] != 0;
        out[i] = keep ? image[((int64_t)c * H + y) * W + x] : 0.0f;
    }
}

int launch_mask_patches(sisic_ctx* ctx, const float* image, const uint8_t* masks, float* out, int S, int C, int H, int W,
                        int patch, hipStream_t s) {
    SISIC_REQUIRE(image && masks && out && S > 0 && patch > 0, "mask_patches: bad arguments");
    const int ph = H / patch, pw = W / patch;
    SISIC_REQUIRE(ph > 0 && pw > 0, "mask_patches: patch %d larger than the %dx%d image", patch, H, W);
    const int64_t total = (int64_t)S * C * H * W;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(mask_patches_kernel, dim3(blocks), dim3(256), 0, s, image, masks, out, C, H, W, patch, ph, pw, total);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

}  // namespace sisic
