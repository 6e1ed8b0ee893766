// classifier_bwd.hip -- gradient of the per-class score with respect to the classifier INPUT (SURVEY.md section 8f
// rank 3): what captum's IntegratedGradients and the plain-gradient fallback of xai/XAI.py:1039-1109 differentiate,
//     score(x) = log(softmax(resnet18(preprocess(x)))[c] + 1e-8)          (XAI.py:443-459)
// The convolutions of the backward pass are convolutions too and run on the forward kernels (conv_mfma.hip /
// conv_winograd.hip) with transposed, tap-flipped filters -- stride 2 through the zero-insertion input mode --
// so this file only holds what is not a convolution:
//   * score_head_bwd_kernel : d score / d (last activation): softmax/log, Linear, global average pool, ReLU mask
//   * relu_bwd_kernel       : dy * [y > 0]
//   * scatter_add_even      : the input side of a transposed 1x1 stride-2 convolution (downsample branch)
//   * maxpool_bwd_kernel    : MaxPool2d(3, 2, 1) backward with PyTorch's first-maximum tie rule, fused ReLU mask
//   * stem_bwd_kernel       : transposed 7x7 stride-2 convolution of the stem (64 -> 3 channels at 224x224)
//   * preprocess_bwd_kernel : adjoint of normalise . bilinear(align_corners=False) . clamp((x+1)/2, 0, 1)
#include "common.h"

namespace sisic {

// g[b,c,p] = [act > 0] * (1/HW) * sum_k W[k,c] * dscore/dlogit_k,   dscore/dlogit_k = p_t/(p_t + 1e-8) * (delta_kt - p_k)
__global__ void __launch_bounds__(256)
score_head_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ fc_w, const float* __restrict__ act,
                      float* __restrict__ g, int C, int HW, int n_classes, int target, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % C);
        const int64_t b = i / ((int64_t)HW * C);
        const float* lg = logits + b * n_classes;
        float m = lg[0];
        for (int k = 1; k < n_classes; ++k) m = fmaxf(m, lg[k]);
        float sum = 0.0f;
        for (int k = 0; k < n_classes; ++k) sum += expf(lg[k] - m);
        const float pt = expf(lg[target] - m) / sum;
        const float coef = pt / (pt + 1e-8f);
        float d = 0.0f;
        for (int k = 0; k < n_classes; ++k) {
            const float pk = expf(lg[k] - m) / sum;
            d += fc_w[(size_t)k * C + c] * (coef * ((k == target ? 1.0f : 0.0f) - pk));
        }
        g[i] = act[i] > 0.0f ? d / (float)HW : 0.0f;
    }
}

int launch_score_head_bwd(sisic_ctx* ctx, const float* logits, const float* fc_w, const float* act, float* g, int B, int C,
                          int HW, int n_classes, int target, hipStream_t s) {
    SISIC_REQUIRE(logits && fc_w && act && g && target >= 0 && target < n_classes, "score_head_bwd: bad arguments");
    const int64_t total = (int64_t)B * C * HW;
    ProfileScope prof(ctx, s, PK_OTHER, 8.0 * total, 0.0);
    hipLaunchKernelGGL(score_head_bwd_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, s,
                       logits, fc_w, act, g, C, HW, n_classes, target, total);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

__global__ void __launch_bounds__(256)
relu_bwd_kernel(const float* dy, const float* y, float* out, int64_t n) {     // out may alias dy
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = y[i] > 0.0f ? dy[i] : 0.0f;
}

int launch_relu_bwd(sisic_ctx* ctx, const float* dy, const float* y, float* out, int64_t n, hipStream_t s) {
    SISIC_REQUIRE(dy && y && out && n > 0, "relu_bwd: bad arguments");
    ProfileScope prof(ctx, s, PK_OTHER, 12.0 * n, 0.0);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 16384)), dim3(256), 0, s, dy, y, out, n);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// dst[b,c,2i,2j] += src[b,c,i,j]   (dst is [planes,H,W], src [planes,OH,OW] with OH = (H-1)/2+1)
__global__ void __launch_bounds__(256)
scatter_add_even_kernel(float* __restrict__ dst, const float* __restrict__ src, int H, int W, int OH, int OW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % OW);
        const int64_t r = i / OW;
        const int ii = (int)(r % OH);
        const int64_t plane = r / OH;
        dst[(plane * H + 2 * ii) * W + 2 * j] += src[i];
    }
}

int launch_scatter_add_even(sisic_ctx* ctx, float* dst, const float* src, int planes, int H, int W, hipStream_t s) {
    SISIC_REQUIRE(dst && src && planes > 0 && H > 0 && W > 0, "scatter_add_even: bad arguments");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)planes * OH * OW;
    ProfileScope prof(ctx, s, PK_OTHER, 12.0 * total, 0.0);
    hipLaunchKernelGGL(scatter_add_even_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 16384)), dim3(256), 0, s,
                       dst, src, H, W, OH, OW, total);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// MaxPool2d(3, 2, 1) backward in gather form, then the ReLU mask of the pooled tensor's producer:
//   dx[y,x] = [x_in[y,x] > 0] * sum over the (up to four) windows that contain (y,x) and whose FIRST maximum in
//   row-major scan order is at (y,x) of dm[window]         (PyTorch's max_pool2d tie rule)
__global__ void __launch_bounds__(256)
maxpool_bwd_kernel(const float* __restrict__ dm, const float* __restrict__ xin, float* __restrict__ dx, int H, int W,
                   int OH, int OW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int64_t r = i / W;
        const int y = (int)(r % H);
        const int64_t plane = r / H;
        const float* src = xin + plane * (int64_t)H * W;
        const float v = src[(int64_t)y * W + x];
        float acc = 0.0f;
        if (v > 0.0f) {
            // windows (oy, ox) with 2*oy - 1 <= y <= 2*oy + 1
            for (int oy = (y + 1) / 2 - ((y + 1) % 2 == 0 ? 1 : 0); oy <= (y + 1) / 2; ++oy) {
                if (oy < 0 || oy >= OH) continue;
                for (int ox = (x + 1) / 2 - ((x + 1) % 2 == 0 ? 1 : 0); ox <= (x + 1) / 2; ++ox) {
                    if (ox < 0 || ox >= OW) continue;
                    // first maximum of the window in scan order
                    float best = -INFINITY;
                    int by = -1, bx = -1;
                    for (int ky = 0; ky < 3; ++ky) {
                        const int yy = 2 * oy - 1 + ky;
                        if (yy < 0 || yy >= H) continue;
                        for (int kx = 0; kx < 3; ++kx) {
                            const int xx = 2 * ox - 1 + kx;
                            if (xx < 0 || xx >= W) continue;
                            const float t = src[(int64_t)yy * W + xx];
                            if (t > best) { best = t; by = yy; bx = xx; }
                        }
                    }
                    if (by == y && bx == x) acc += dm[(plane * OH + oy) * (int64_t)OW + ox];
                }
            }
        }
        dx[i] = acc;
    }
}

int launch_maxpool_bwd(sisic_ctx* ctx, const float* dm, const float* xin, float* dx, int planes, int H, int W, hipStream_t s) {
    SISIC_REQUIRE(dm && xin && dx && planes > 0, "maxpool_bwd: bad arguments");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)planes * H * W;
    ProfileScope prof(ctx, s, PK_OTHER, 8.0 * total, 0.0);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 32768)), dim3(256), 0, s, dm,
                       xin, dx, H, W, OH, OW, total);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// Transposed 7x7 stride-2 padding-3 convolution, 64 -> 3 channels:
//   dp[b,ci,Y,X] = sum_{co,ky,kx} W[co,ci,ky,kx] * g[b,co,(Y+3-ky)/2,(X+3-kx)/2]   over the taps for which both
//   quotients are integers inside the map.  W = the BN-folded OIHW stem filter.
// A workgroup owns a 32x32 output tile of one image and walks the 64 channels in chunks of 8: the 19x19 patch of g
// the tile can reach and the chunk's filters ([co][ky][kx][ci(4)]: the three input channels of a tap are one
// ds_read_b128) sit in LDS; a thread produces a 2x2 block of pixels, i.e. all four tap parities, so every g value it
// reads is used by four pixels.
constexpr int SB_T = 32, SB_G = SB_T / 2 + 3, SB_CC = 8;       // output tile, g patch edge (19), channels per chunk

__global__ void __launch_bounds__(256)
stem_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dp, int CO, int OH, int OW,
                int H, int W, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float wl[SB_CC * 49 * 4];
    __shared__ float gl[SB_CC * SB_G * SB_G];
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int Y0 = ty * SB_T, X0 = tx * SB_T;
    const int oy_lo = Y0 / 2 - 1, ox_lo = X0 / 2 - 1;           // first g row / column any pixel of the tile can reach
    const int tid = threadIdx.x;
    const int py = tid / 16, px = tid % 16;                        // 2x2 pixel block (2*py .. 2*py+1, 2*px .. 2*px+1)
    float acc[2][2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[i][j][c] = 0.0f;
    for (int c0 = 0; c0 < CO; c0 += SB_CC) {
        __syncthreads();
        for (int i = tid; i < SB_CC * 49; i += 256) {
            const int co = c0 + i / 49, tap = i % 49;
            const bool v = co < CO;
            wl[i * 4 + 0] = v ? w[((size_t)co * 3 + 0) * 49 + tap] : 0.0f;
            wl[i * 4 + 1] = v ? w[((size_t)co * 3 + 1) * 49 + tap] : 0.0f;
            wl[i * 4 + 2] = v ? w[((size_t)co * 3 + 2) * 49 + tap] : 0.0f;
            wl[i * 4 + 3] = 0.0f;
        }
        for (int i = tid; i < SB_CC * SB_G * SB_G; i += 256) {
            const int cc = i / (SB_G * SB_G), r = i % (SB_G * SB_G);
            const int oy = oy_lo + r / SB_G, ox = ox_lo + r % SB_G;
            const bool v = c0 + cc < CO && oy >= 0 && oy < OH && ox >= 0 && ox < OW;
            gl[i] = v ? g[(((size_t)b * CO + c0 + cc) * OH + oy) * OW + ox] : 0.0f;
        }
        __syncthreads();
        // pixel (2py+i, 2px+j) of the tile: Y = Y0 + 2py + i; taps ky with (Y+3-ky) even: ky = (i+1)%2, +2, ...;
        // g row oy = (Y + 3 - ky) / 2 = Y0/2 + py + (i + 3 - ky) / 2   ->  patch row py + 1 + (i + 3 - ky) / 2  in [0, 18]
        for (int cc = 0; cc < SB_CC; ++cc) {
            const float* gp = gl + cc * SB_G * SB_G;
            const float* wp = wl + cc * 49 * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int ky = (i + 1) & 1; ky < 7; ky += 2) {
                    const int gr = py + 1 + (i + 3 - ky) / 2;      // (i + 3 - ky) is even and in [-2, 4]
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int kx = (j + 1) & 1; kx < 7; kx += 2) {
                            const int gc = px + 1 + (j + 3 - kx) / 2;
                            const float gv = gp[gr * SB_G + gc];
                            const float4 wv = *reinterpret_cast<const float4*>(&wp[(ky * 7 + kx) * 4]);
                            acc[i][j][0] += gv * wv.x;
                            acc[i][j][1] += gv * wv.y;
                            acc[i][j][2] += gv * wv.z;
                        }
                    }
                }
            }
        }
    }
    const size_t HWp = (size_t)H * W;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int Y = Y0 + 2 * py + i, X = X0 + 2 * px + j;
            if (Y < H && X < W) {
                float* o = dp + (size_t)b * 3 * HWp + (size_t)Y * W + X;
                o[0] = acc[i][j][0];
                o[HWp] = acc[i][j][1];
                o[2 * HWp] = acc[i][j][2];
            }
        }
}

int launch_stem_bwd(sisic_ctx* ctx, const float* g, const float* w_oihw, float* dp, int B, int CO, int OH, int OW, int H,
                    int W, hipStream_t s) {
    SISIC_REQUIRE(g && w_oihw && dp && B > 0 && CO > 0, "stem_bwd: bad arguments");
    SISIC_REQUIRE(OH == (H + 6 - 7) / 2 + 1 && OW == (W + 6 - 7) / 2 + 1, "stem_bwd: %dx%d is not the stem output of %dx%d", OH, OW, H, W);
    const int tiles_x = (W + SB_T - 1) / SB_T, tiles_y = (H + SB_T - 1) / SB_T;
    const int64_t nwg = (int64_t)B * tiles_x * tiles_y;
    SISIC_REQUIRE(nwg < (int64_t(1) << 31), "stem_bwd: grid too large");
    ProfileScope prof(ctx, s, PK_OTHER, 4.0 * B * ((double)CO * OH * OW + 3.0 * H * W), 2.0 * B * 3.0 * CO * 49.0 * OH * OW);
    hipLaunchKernelGGL(stem_bwd_kernel, dim3((unsigned)nwg), dim3(256), 0, s, g, w_oihw, dp, CO, OH, OW, H, W, tiles_x, tiles_y);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// Adjoint of preprocess_kernel (classifier.hip):  p = (bilinear(clamp((x+1)/2, 0, 1)) - mean) / std.
//   dx[b,c,y,x] = 0.5 * [0 <= (x+1)/2 <= 1] / std_c * sum_{Y,X} wy(Y,y) * wx(X,x) * dp[b,c,Y,X]
// with the forward's own source-index arithmetic recomputed per target row/column, so every edge case (clamped source
// coordinate at the top/left, duplicated last row/column) is the forward's by construction.
__global__ void __launch_bounds__(256)
preprocess_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ x, float* __restrict__ dx, int C, int H, int W,
                      int OH, int OW, float sh, float sw, float is0, float is1, float is2, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xs = (int)(i % W);
        int64_t r = i / W;
        const int ys = (int)(r % H);
        r /= H;
        const int c = (int)(r % C);
        const int64_t b = r / C;
        const float u = (x[i] + 1.0f) / 2.0f;
        float acc = 0.0f;
        if (u >= 0.0f && u <= 1.0f) {
            const float* src = dp + (b * C + c) * (int64_t)OH * OW;
            if (OH == H && OW == W) {
                acc = src[(int64_t)ys * W + xs];
            } else {
                // target rows whose two source rows can include ys: source coordinate in (ys - 1, ys + 1)
                const int Ylo = max(0, (int)floorf(((float)ys - 0.5f) / sh - 0.5f) - 1);
                const int Yhi = min(OH - 1, (int)ceilf(((float)ys + 1.5f) / sh - 0.5f) + 1);
                const int Xlo = max(0, (int)floorf(((float)xs - 0.5f) / sw - 0.5f) - 1);
                const int Xhi = min(OW - 1, (int)ceilf(((float)xs + 1.5f) / sw - 0.5f) + 1);
                for (int Y = Ylo; Y <= Yhi; ++Y) {
                    const float fy = fmaxf(sh * ((float)Y + 0.5f) - 0.5f, 0.0f);
                    const int y0 = min((int)fy, H - 1), y1 = min(y0 + 1, H - 1);
                    const float ly = fy - (float)y0;
                    const float wy = (y0 == ys ? 1.0f - ly : 0.0f) + (y1 == ys ? ly : 0.0f);
                    if (wy == 0.0f) continue;
                    float row = 0.0f;
                    for (int X = Xlo; X <= Xhi; ++X) {
                        const float fx = fmaxf(sw * ((float)X + 0.5f) - 0.5f, 0.0f);
                        const int x0 = min((int)fx, W - 1), x1 = min(x0 + 1, W - 1);
                        const float lx = fx - (float)x0;
                        const float wx = (x0 == xs ? 1.0f - lx : 0.0f) + (x1 == xs ? lx : 0.0f);
                        if (wx != 0.0f) row += wx * src[(int64_t)Y * OW + X];
                    }
                    acc += wy * row;
                }
            }
            acc *= 0.5f * (c == 0 ? is0 : (c == 1 ? is1 : is2));
        }
        dx[i] = acc;
    }
}

int launch_preprocess_bwd(sisic_ctx* ctx, const float* dp, const float* x, float* dx, int B, int H, int W, int OH, int OW,
                          hipStream_t s) {
    SISIC_REQUIRE(dp && x && dx && B > 0 && H <= OH && W <= OW, "preprocess_bwd: bad arguments");
    const int64_t total = (int64_t)B * 3 * H * W;
    ProfileScope prof(ctx, s, PK_OTHER, 4.0 * B * 3 * ((double)2 * H * W + (double)OH * OW), 0.0);
    hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, s, dp, x,
                       dx, 3, H, W, OH, OW, (float)H / (float)OH, (float)W / (float)OW, 1.0f / 0.229f, 1.0f / 0.224f,
                       1.0f / 0.225f, total);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

}  // namespace sisic

namespace sisic {

// out = relu(y + identity)   (the last block's tail when its conv2 output is kept for Grad-CAM)
__global__ void __launch_bounds__(256)
add_relu_kernel(const float* __restrict__ y, const float* __restrict__ identity, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = fmaxf(y[i] + identity[i], 0.0f);
}

int launch_add_relu(sisic_ctx* ctx, const float* y, const float* identity, float* out, int64_t n, hipStream_t s) {
    SISIC_REQUIRE(y && identity && out && n > 0, "add_relu: bad arguments");
    ProfileScope prof(ctx, s, PK_OTHER, 12.0 * n, 0.0);
    hipLaunchKernelGGL(add_relu_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 16384)), dim3(256), 0, s, y, identity, out, n);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

// Grad-CAM (Selvaraju et al.) on layer4[-1].conv2 for the raw class logit, as pytorch_grad_cam's GradCAM computes it
// (xai/XAI.py:2945-3035: target_layers=[model.layer4[-1].conv2], ClassifierOutputTarget(c)):
//   A = conv2 output BEFORE its BatchNorm, G = d logit_c / d A, alpha_k = mean_hw G_k,
//   cam = relu(sum_k alpha_k A_k) -> (cam - min) / (1e-7 + max) -> bilinear to SxS -> (.. - min) / (1e-7 + max) again.
// With BatchNorm folded, Y = s A + b is what the convolution produced and d logit / d Y = W_fc[c,k] / HW * [out > 0], so
//   alpha_k A_k = (W_fc[c,k] / HW) * frac_k * (Y_k - b_k),   frac_k = share of positive outputs in channel k
// (the BatchNorm scale s_k cancels).  One workgroup per image.
__global__ void __launch_bounds__(256)
gradcam_kernel(const float* __restrict__ y, const float* __restrict__ outp, const float* __restrict__ fc_w,
               const float* __restrict__ bias, float* __restrict__ cam, int C, int h, int w, int S, int target) {
    extern __shared__ float sm[];
    float* coef = sm;                  // [C]
    float* cam_lo = sm + C;            // [h*w]
    float* red = cam_lo + h * w;       // [512] reduction scratch (min, max)
    const int b = blockIdx.x, tid = threadIdx.x, HW = h * w;
    const float* yb = y + (size_t)b * C * HW;
    const float* ob = outp + (size_t)b * C * HW;
    for (int k = tid; k < C; k += blockDim.x) {
        int pos = 0;
        for (int p = 0; p < HW; ++p) pos += ob[(size_t)k * HW + p] > 0.0f ? 1 : 0;
        coef[k] = fc_w[(size_t)target * C + k] / (float)HW * ((float)pos / (float)HW);
    }
    __syncthreads();
    for (int p = tid; p < HW; p += blockDim.x) {
        float a = 0.0f;
        for (int k = 0; k < C; ++k) a += coef[k] * (yb[(size_t)k * HW + p] - bias[k]);
        cam_lo[p] = fmaxf(a, 0.0f);
    }
    __syncthreads();
    auto block_minmax = [&](float vmin, float vmax, float* omin, float* omax) {
        red[tid] = vmin;
        red[256 + tid] = vmax;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) {
                red[tid] = fminf(red[tid], red[tid + off]);
                red[256 + tid] = fmaxf(red[256 + tid], red[256 + tid + off]);
            }
            __syncthreads();
        }
        *omin = red[0];
        *omax = red[256];
        __syncthreads();
    };
    float mn = INFINITY, mx = -INFINITY;
    for (int p = tid; p < HW; p += blockDim.x) { mn = fminf(mn, cam_lo[p]); mx = fmaxf(mx, cam_lo[p]); }
    float lo_min, lo_max;
    block_minmax(mn, mx, &lo_min, &lo_max);
    for (int p = tid; p < HW; p += blockDim.x) cam_lo[p] = (cam_lo[p] - lo_min) / (1e-7f + (lo_max - lo_min));
    __syncthreads();
    // bilinear, half-pixel centres, clamped source index (cv2.INTER_LINEAR / align_corners=False when upscaling)
    float* cb = cam + (size_t)b * S * S;
    const float sh = (float)h / (float)S, sw = (float)w / (float)S;
    mn = INFINITY; mx = -INFINITY;
    for (int i = tid; i < S * S; i += blockDim.x) {
        const int Y = i / S, X = i % S;
        const float fy = fmaxf(sh * ((float)Y + 0.5f) - 0.5f, 0.0f), fx = fmaxf(sw * ((float)X + 0.5f) - 0.5f, 0.0f);
        const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float top = cam_lo[y0 * w + x0] * (1.0f - lx) + cam_lo[y0 * w + x1] * lx;
        const float bot = cam_lo[y1 * w + x0] * (1.0f - lx) + cam_lo[y1 * w + x1] * lx;
        const float v = top * (1.0f - ly) + bot * ly;
        cb[i] = v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    float hi_min, hi_max;
    block_minmax(mn, mx, &hi_min, &hi_max);
    for (int i = tid; i < S * S; i += blockDim.x) cb[i] = (cb[i] - hi_min) / (1e-7f + (hi_max - hi_min));
}

int launch_gradcam(sisic_ctx* ctx, const float* y, const float* outp, const float* fc_w, const float* bias, float* cam, int B,
                   int C, int h, int w, int S, int target, hipStream_t s) {
    SISIC_REQUIRE(y && outp && fc_w && bias && cam && B > 0 && C > 0 && h > 0 && w > 0 && S > 0, "gradcam: bad arguments");
    const size_t lds = ((size_t)C + (size_t)h * w + 512) * sizeof(float);
    SISIC_REQUIRE(lds <= 64 * 1024, "gradcam: %d channels x %dx%d do not fit the scratch", C, h, w);
    ProfileScope prof(ctx, s, PK_OTHER, 8.0 * B * C * h * w + 4.0 * B * S * S, 0.0);
    hipLaunchKernelGGL(gradcam_kernel, dim3(B), dim3(256), lds, s, y, outp, fc_w, bias, cam, C, h, w, S, target);
    SISIC_HIP(hipGetLastError());
    return SISIC_OK;
}

}  // namespace sisic
