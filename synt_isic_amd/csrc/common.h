// Internal declarations shared by the libsisic_hip.so translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/sisic.h"

namespace sisic {

void set_error(const char* fmt, ...);

#define SISIC_HIP(call)                                                                   \
    do {                                                                                  \
        hipError_t _e = (call);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::sisic::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e),     \
                               __FILE__, __LINE__);                                       \
            return SISIC_EHIP;                                                            \
        }                                                                                 \
    } while (0)

#define SISIC_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            ::sisic::set_error(__VA_ARGS__);     \
            return SISIC_EINVAL;                 \
        }                                        \
    } while (0)

#define SISIC_TRY(expr)             \
    do {                            \
        int _rc = (expr);           \
        if (_rc != SISIC_OK) return _rc; \
    } while (0)

// PK_WINO_MAIN / PK_WINO_BF3: the Winograd kernel families on their own -- the f32-MFMA forms (tile_cfg 66, 68-73, 78, 79) and
// the bf16x3 form (tile_cfg 74); their launches are credited to PK_CONV3 (the class) AND to their slot (same event pair)
enum ProfileKind { PK_CONV3 = 0, PK_CONV1 = 1, PK_GN = 2, PK_ATTN = 3, PK_DDPM = 4, PK_OTHER = 5, PK_WINO_MAIN = 6, PK_WINO_BF3 = 7, PK_COUNT = 8 };

struct ProfileSlot {
    double ms = 0, bytes = 0, flops = 0, flops_exec = 0;   // flops: algorithmic (2*MAC of the direct form); flops_exec: issued to the matrix pipe
    int64_t launches = 0;
};

struct PendingEvent {
    hipEvent_t start, stop;
    int kind;
    int kind2;      // second slot credited with the same launch, or -1
};

}  // namespace sisic

struct sisic_ctx {
    int device = 0;
    int num_cus = 0;
    bool profiling = false;
    sisic::ProfileSlot prof[sisic::PK_COUNT];
    std::vector<sisic::PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
    // partial outputs of K-split convolutions (conv_winograd.hip): one scratch buffer per stream, grown on demand, so
    // that models sampling concurrently on different streams of this context never share it
    struct SplitK { float* p = nullptr; size_t floats = 0; };
    std::map<hipStream_t, SplitK> splitk;
    std::mutex splitk_mutex;
    std::atomic<uint64_t> scratch_generation{0};    // bumped whenever a scratch buffer is re-allocated: captured graphs that
                                                    // may hold its old address are rebuilt
};

namespace sisic {

// RAII-less helper: brackets one launch with events when profiling is on.
struct ProfileScope {
    sisic_ctx* ctx;
    hipStream_t stream;
    PendingEvent ev{};
    bool active = false;
    ProfileScope(sisic_ctx* c, hipStream_t s, int kind, double bytes, double flops, double flops_exec = -1.0, int kind2 = -1);
    ~ProfileScope();
};

int profile_collect(sisic_ctx* ctx);

// Opt a kernel in to more than 64 KB of dynamic LDS.  The attribute belongs to the (kernel, device) pair, and several
// devices and threads may use one process (sisic_create(device_id)), so every template instance keeps one bit per device
// in an atomic mask; setting the attribute twice from two racing threads is harmless (same value).
inline int ensure_dynamic_lds(sisic_ctx* ctx, const void* kern, int bytes, std::atomic<uint64_t>& done) {
    const uint64_t bit = uint64_t(1) << (ctx->device & 63);
    if (done.load(std::memory_order_acquire) & bit) return SISIC_OK;
    SISIC_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.fetch_or(bit, std::memory_order_release);
    return SISIC_OK;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// Latency mode (tile_cfg 78 / 79): ways the input channels of a Winograd convolution are split over workgroups so that ONE
// image already offers ~256 workgroups.  A function of the layer shape only -- never of the batch -- so that an image's
// bits do not depend on the batch it is computed in.
inline int wino_latency_ksplit(int Cout, int Cin, int Hout, int Wout) {
    const int co_t = Cout > 64 ? 128 : 64;
    const int per_image = cdiv(Hout, 8) * cdiv(Wout, 16) * cdiv(Cout, co_t);
    const int nchunks = cdiv(Cin, 8);
    int k = 1;
    while (per_image * k < 256 && k < 8 && nchunks % (2 * k) == 0) k *= 2;
    return k;
}
// segments a plane is cut into by the K-split plane reduction (one workgroup and one statistics slot each): 1024 pixels
inline int wino_latency_segments(int Hout, int Wout) { return cdiv(Hout * Wout, 1024); }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// conv weight packing geometry (must match conv_mfma.hip)
constexpr int CONV_CO_TILE = 64;
inline int conv_ci_chunk(int ksize) { return ksize == 1 ? 32 : (ksize == 3 ? 16 : 4); }   // channel padding of the packed weights (1x1: tiles use 16 or 32)
inline int conv_cin_pad(int cin, int ksize) { return round_up(cin, conv_ci_chunk(ksize)); }
inline int conv_cout_pad(int cout) { return round_up(cout, CONV_CO_TILE); }
// floats of a packed filter: 1x1 filters carry three layouts (pack_device.h: generic, conv_pointwise.hip's, the bf16x3 split)
inline int64_t conv_packed_floats(int cout, int cin, int ksize) {
    const int64_t first = (int64_t)conv_cin_pad(cin, ksize) * ksize * ksize * conv_cout_pad(cout);
    return ksize == 1 ? 3 * first + first / 2 : first;
}

// launchers implemented in the .hip files
int launch_conv2d(sisic_ctx*, const sisic_conv_args& a, hipStream_t s);
int launch_conv_smallcout(sisic_ctx*, const sisic_conv_args& a, hipStream_t s);
// conv_winograd.hip: F(2x2,3x3) for 3x3 stride-1 convolutions
int launch_score_head_bwd(sisic_ctx*, const float* logits, const float* fc_w, const float* act, float* g, int B, int C,
                          int HW, int n_classes, int target, hipStream_t s);
int launch_relu_bwd(sisic_ctx*, const float* dy, const float* y, float* out, int64_t n, hipStream_t s);
int launch_scatter_add_even(sisic_ctx*, float* dst, const float* src, int planes, int H, int W, hipStream_t s);
int launch_maxpool_bwd(sisic_ctx*, const float* dm, const float* xin, float* dx, int planes, int H, int W, hipStream_t s);
int launch_stem_bwd(sisic_ctx*, const float* g, const float* w_oihw, float* dp, int B, int CO, int OH, int OW, int H, int W,
                    hipStream_t s);
int launch_preprocess_bwd(sisic_ctx*, const float* dp, const float* x, float* dx, int B, int H, int W, int OH, int OW,
                          hipStream_t s);
int launch_add_relu(sisic_ctx*, const float* y, const float* identity, float* out, int64_t n, hipStream_t s);
int launch_gradcam(sisic_ctx*, const float* y, const float* outp, const float* fc_w, const float* bias, float* cam, int B, int C,
                   int h, int w, int S, int target, hipStream_t s);
int conv_stats_slots(const sisic_conv_args& a);
bool conv_finalizes(const sisic_conv_args& a);       // the launch leaves the following GroupNorm's (scale, shift) itself (sisic.h)
// conv_pointwise.hip: the lean 1x1 kernel (tile_cfg 20)
bool conv_pointwise_applicable(const sisic_conv_args& a);
int conv_pointwise_stats_slots(const sisic_conv_args& a);
int launch_conv_pointwise(sisic_ctx*, const sisic_conv_args& a, hipStream_t s);
// conv_pointwise_bf3.hip: the same GEMM with fp32-equivalent products on the bf16 matrix pipe (tile_cfg 28)
bool conv_pointwise_bf3_applicable(const sisic_conv_args& a);
int launch_conv_pointwise_bf3(sisic_ctx*, const sisic_conv_args& a, hipStream_t s);
// mean_rstd (optional, training): [B, groups, 2] = (mean, rstd) of every (sample, group)
int launch_gn_finalize(sisic_ctx*, const float* st0, int c0, int slots0, const float* st1, int c1, int slots1, int B,
                       int HW, int groups, float eps, const float* gamma, const float* beta, float* scale, float* shift,
                       hipStream_t s, float* mean_rstd = nullptr);
int launch_conv_winograd(sisic_ctx*, const sisic_conv_args& a, const float* u_packed, int cfg, hipStream_t s);
int launch_winograd_pack(sisic_ctx*, const float* w, int Cout, int Cin, float* packed, hipStream_t s);
int64_t winograd_packed_numel(int Cout, int Cin);
// floats of the first Winograd layout [Cin_pad][16][cout_pad]; the wide layout follows it in the same buffer
inline int64_t winograd_first_floats(int Cout, int Cin) { return (int64_t)round_up(Cin, 16) * 16 * conv_cout_pad(Cout); }
int launch_conv_pack(sisic_ctx*, const float* w, int Cout, int Cin, int k, float* packed, hipStream_t s);
int launch_gn_stats(sisic_ctx*, const float* in0, int c0, const float* in1, int c1, int B, int HW, int groups,
                    float eps, const float* gamma, const float* beta, float* scale, float* shift, hipStream_t s,
                    float* mean_rstd = nullptr);
int launch_attention(sisic_ctx*, const float* qkv, float* out, int B, int C, int N, int head_dim, hipStream_t s);
int launch_ddpm_step(sisic_ctx*, const float* eps, const float* x, const float* z, float* out, int64_t n,
                     float sb, float sa, float c0, float c1, float sigma, float clip, hipStream_t s);
// graph-replayed sampling loop (elementwise.hip): per-step parameters selected on the device by a step index
size_t loop_state_bytes();     // {int step; int pad; const float* noise_base}
int launch_loop_select_row(sisic_ctx*, const float* table, int R, const void* state, float* out, hipStream_t s);
int launch_loop_advance(sisic_ctx*, void* state, hipStream_t s);
int launch_ddpm_step_indexed(sisic_ctx*, const float* eps, float* x, int64_t n, const void* state, const float* coef,
                             const int* zrow, float clip, hipStream_t s);
int launch_denorm_u8(sisic_ctx*, const float* x, uint8_t* out, int B, int C, int H, int W, hipStream_t s, int form = 0);
// time embedding: sinusoid -> linear1 -> SiLU -> linear2 -> SiLU  (weights transposed [in][out])
// save_* (optional, training): the sinusoid [B, 2 n_freqs] and the two linear outputs before their SiLU [B, hidden]
int launch_temb_mlp(sisic_ctx*, const float* t_vals, int B, const float* freqs, int n_freqs, const float* w1t,
                    const float* b1, const float* w2t, const float* b2, int hidden, float* temb_act, hipStream_t s,
                    float* save_emb = nullptr, float* save_h1 = nullptr, float* save_t2 = nullptr);
// out[b, r] = sum_k wt[k][r] * x[b][k] + bias[r]
int launch_linear_t(sisic_ctx*, const float* x, int B, int K, const float* wt, const float* bias, int R,
                    float* out, hipStream_t s);
int launch_transpose2d(sisic_ctx*, const float* in, int rows, int cols, float* out, int out_ld, int out_col0,
                       hipStream_t s);
// classifier.hip
int launch_preprocess(sisic_ctx*, const float* x, float* out, int B, int H, int W, int OH, int OW, hipStream_t s);
int launch_maxpool(sisic_ctx*, const float* x, float* out, int B, int C, int H, int W, hipStream_t s);
int launch_avgpool_fc(sisic_ctx*, const float* x, const float* w, const float* bias, float* out, int B, int C, int HW,
                      int n_out, hipStream_t s);
int launch_class_scores(sisic_ctx*, const float* logits, int B, int n, int target, float* prob, float* logscore,
                        hipStream_t s);
int launch_mask_patches(sisic_ctx*, const float* image, const uint8_t* masks, float* out, int S, int C, int H, int W,
                        int patch, hipStream_t s);

}  // namespace sisic
