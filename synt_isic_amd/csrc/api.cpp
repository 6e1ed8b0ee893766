// api.cpp -- context, error reporting, launch timing and the extern "C" operator entry points
// of libsisic_hip.so (include/sisic.h).
#include <cstring>

#include "common.h"

namespace sisic {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- launch timing ----------------------------------------------------------------------
// bench.py's roofline leg: every launch is bracketed by two HIP events recorded on the stream
// the kernel runs on; elapsed times are accumulated per kernel class when the profile is read.
static hipEvent_t take_event(sisic_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfileScope::ProfileScope(sisic_ctx* c, hipStream_t s, int kind, double bytes, double flops, double flops_exec, int kind2)
    : ctx(c), stream(s) {
    if (!c || !c->profiling) return;
    ev.start = take_event(c);
    ev.stop = take_event(c);
    ev.kind = kind;
    ev.kind2 = kind2;
    if (!ev.start || !ev.stop) return;
    for (int k : {kind, kind2}) {
        if (k < 0) continue;
        c->prof[k].bytes += bytes;
        c->prof[k].flops += flops;
        c->prof[k].flops_exec += (flops_exec < 0.0) ? flops : flops_exec;
        c->prof[k].launches += 1;
    }
    (void)hipEventRecord(ev.start, s);
    active = true;
}

ProfileScope::~ProfileScope() {
    if (!active) return;
    (void)hipEventRecord(ev.stop, stream);
    ctx->pending.push_back(ev);
}

int profile_collect(sisic_ctx* ctx) {
    for (auto& ev : ctx->pending) {
        SISIC_HIP(hipEventSynchronize(ev.stop));
        float ms = 0.f;
        SISIC_HIP(hipEventElapsedTime(&ms, ev.start, ev.stop));
        ctx->prof[ev.kind].ms += ms;
        if (ev.kind2 >= 0) ctx->prof[ev.kind2].ms += ms;
        ctx->event_pool.push_back(ev.start);
        ctx->event_pool.push_back(ev.stop);
    }
    ctx->pending.clear();
    return SISIC_OK;
}

}  // namespace sisic

using namespace sisic;

extern "C" {

int sisic_abi_version(void) { return SISIC_ABI_VERSION; }

const char* sisic_last_error(void) { return sisic::g_err; }

int sisic_create(int device_id, sisic_ctx** out) {
    SISIC_REQUIRE(out != nullptr, "sisic_create: out is NULL");
    *out = nullptr;
    int n = 0;
    SISIC_HIP(hipGetDeviceCount(&n));
    SISIC_REQUIRE(device_id >= 0 && device_id < n, "sisic_create: device %d of %d", device_id, n);
    SISIC_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    SISIC_HIP(hipGetDeviceProperties(&prop, device_id));
    SISIC_REQUIRE(std::strncmp(prop.gcnArchName, "gfx950", 6) == 0,
                  "sisic_create: device %d is %s; this library is built for gfx950 (MI355X) only", device_id,
                  prop.gcnArchName);
    auto* ctx = new sisic_ctx();
    ctx->device = device_id;
    ctx->num_cus = prop.multiProcessorCount;
    *out = ctx;
    return SISIC_OK;
}

int sisic_destroy(sisic_ctx* ctx) {
    if (!ctx) return SISIC_OK;
    (void)profile_collect(ctx);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (auto& kv : ctx->splitk)
        if (kv.second.p) (void)hipFree(kv.second.p);
    delete ctx;
    return SISIC_OK;
}

int64_t sisic_conv_packed_numel(int Cout, int Cin, int ksize) {
    if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3 && ksize != 7)) return -1;
    return conv_packed_floats(Cout, Cin, ksize);
}

int sisic_conv_pack_weights(sisic_ctx* ctx, const float* w, int Cout, int Cin, int ksize, float* packed, void* stream) {
    SISIC_REQUIRE(ctx && w && packed && Cout > 0 && Cin > 0, "conv_pack_weights: bad arguments");
    return launch_conv_pack(ctx, w, Cout, Cin, ksize, packed, static_cast<hipStream_t>(stream));
}

int64_t sisic_conv_winograd_numel(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return -1;
    return winograd_packed_numel(Cout, Cin);
}

int sisic_conv_winograd_pack(sisic_ctx* ctx, const float* w, int Cout, int Cin, float* packed, void* stream) {
    SISIC_REQUIRE(ctx && w && packed && Cout > 0 && Cin > 0, "conv_winograd_pack: bad arguments");
    return launch_winograd_pack(ctx, w, Cout, Cin, packed, static_cast<hipStream_t>(stream));
}

int sisic_conv2d(sisic_ctx* ctx, const sisic_conv_args* args, void* stream) {
    SISIC_REQUIRE(ctx && args, "conv2d: null argument");
    return launch_conv2d(ctx, *args, static_cast<hipStream_t>(stream));
}

int sisic_conv_stats_slots(const sisic_conv_args* args) { return args ? conv_stats_slots(*args) : 0; }
int sisic_conv_finalizes(const sisic_conv_args* args) { return args && conv_finalizes(*args) ? 1 : 0; }

int sisic_groupnorm_finalize(sisic_ctx* ctx, const float* stats0, int c0, int slots0, const float* stats1, int c1,
                             int slots1, int B, int HW, int groups, float eps, const float* gamma, const float* beta,
                             float* scale, float* shift, void* stream) {
    SISIC_REQUIRE(ctx, "groupnorm_finalize: null context");
    return launch_gn_finalize(ctx, stats0, c0, slots0, stats1, c1, slots1, B, HW, groups, eps, gamma, beta, scale,
                              shift, static_cast<hipStream_t>(stream));
}

int sisic_groupnorm_stats(sisic_ctx* ctx, const float* in0, int c0, const float* in1, int c1, int B, int HW,
                          int groups, float eps, const float* gamma, const float* beta, float* scale, float* shift,
                          void* stream) {
    SISIC_REQUIRE(ctx, "groupnorm_stats: null context");
    return launch_gn_stats(ctx, in0, c0, in1, c1, B, HW, groups, eps, gamma, beta, scale, shift,
                           static_cast<hipStream_t>(stream));
}

int sisic_attention(sisic_ctx* ctx, const float* qkv, float* out, int B, int C, int N, int head_dim, void* stream) {
    SISIC_REQUIRE(ctx, "attention: null context");
    return launch_attention(ctx, qkv, out, B, C, N, head_dim, static_cast<hipStream_t>(stream));
}

int sisic_ddpm_step(sisic_ctx* ctx, const float* eps, const float* x, const float* z, float* out, int64_t n,
                    float sqrt_beta_prod, float sqrt_alpha_prod, float c0, float c1, float sigma, float clip,
                    void* stream) {
    SISIC_REQUIRE(ctx, "ddpm_step: null context");
    return launch_ddpm_step(ctx, eps, x, z, out, n, sqrt_beta_prod, sqrt_alpha_prod, c0, c1, sigma, clip,
                            static_cast<hipStream_t>(stream));
}

int sisic_denorm_u8(sisic_ctx* ctx, const float* x, uint8_t* out, int B, int C, int H, int W, void* stream) {
    SISIC_REQUIRE(ctx, "denorm_u8: null context");
    return launch_denorm_u8(ctx, x, out, B, C, H, W, static_cast<hipStream_t>(stream));
}

int sisic_denorm_u8_form(sisic_ctx* ctx, const float* x, uint8_t* out, int B, int C, int H, int W, int form, void* stream) {
    SISIC_REQUIRE(ctx, "denorm_u8_form: null context");
    return launch_denorm_u8(ctx, x, out, B, C, H, W, static_cast<hipStream_t>(stream), form);
}

int sisic_profile_enable(sisic_ctx* ctx, int on) {
    SISIC_REQUIRE(ctx, "profile_enable: null context");
    ctx->profiling = on != 0;
    return SISIC_OK;
}

int sisic_profile_read(sisic_ctx* ctx, int kind, double* ms, int64_t* launches, double* bytes, double* flops,
                       double* flops_executed) {
    SISIC_REQUIRE(ctx && kind >= 0 && kind < PK_COUNT, "profile_read: bad arguments");
    SISIC_TRY(profile_collect(ctx));
    if (ms) *ms = ctx->prof[kind].ms;
    if (launches) *launches = ctx->prof[kind].launches;
    if (bytes) *bytes = ctx->prof[kind].bytes;
    if (flops) *flops = ctx->prof[kind].flops;
    if (flops_executed) *flops_executed = ctx->prof[kind].flops_exec;
    return SISIC_OK;
}

int sisic_profile_reset(sisic_ctx* ctx) {
    SISIC_REQUIRE(ctx, "profile_reset: null context");
    SISIC_TRY(profile_collect(ctx));
    for (auto& p : ctx->prof) p = ProfileSlot();
    return SISIC_OK;
}

}  // extern "C"
