// unet_internal.h -- the UNet2DModel instance behind a sisic_unet handle: architecture tables, weight arena, activation
// pool.  Shared by unet.cpp (inference executor, sampling loop) and train.cpp (tape-recording forward, backward, Adam).
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

namespace sisic {

struct ConvW {            // one convolution's parameters
    int cout = 0, cin = 0, k = 0;
    int w_idx = -1, b_idx = -1;   // indices into the state-dict tensor table (raw)
    float* packed = nullptr;      // device, packed layout
    float* wino = nullptr;        // device, Winograd-domain filters (3x3 stride-1 convolutions only)
    bool strided = false;         // stride-2 downsampler: no Winograd form
    const float* bias = nullptr;  // device
    // training only (train.cpp): filters of the backward-data convolution, W'[ci][co][k*k-1-t] = W[co][ci][t]
    float* raw_t = nullptr;       // OIHW of the transposed, tap-flipped filter
    float* packed_t = nullptr;    // its packed layout
    float* wino_t = nullptr;      // its Winograd-domain form (3x3 stride-1 convolutions)
};

struct NormW {
    int c = 0;
    int w_idx = -1, b_idx = -1;
    const float* gamma = nullptr;
    const float* beta = nullptr;
};

struct ResnetW {
    int cin = 0, cout = 0;
    NormW norm1, norm2;
    ConvW conv1, conv2, shortcut;   // shortcut.k == 0 when absent
    int temb_w_idx = -1, temb_b_idx = -1;
    int temb_off = 0;               // column offset in the fused time-embedding projection
};

struct AttnW {
    int c = 0;
    NormW norm;
    int q_w = -1, q_b = -1, k_w = -1, k_b = -1, v_w = -1, v_b = -1, o_w = -1, o_b = -1;
    float* qkv_cat = nullptr;       // [3C, C] raw q|k|v rows (source of the packed form)
    float* qkv_packed = nullptr;    // [3C <- C] 1x1 conv
    float* qkv_bias = nullptr;      // [3C]
    float* qkv_raw_t = nullptr;     // training: [C <- 3C] transposed for the backward-data 1x1 convolution
    float* qkv_packed_t = nullptr;
    ConvW out;                      // to_out.0 as 1x1 conv
};

struct Buf {
    float* p = nullptr;
    int C = 0, H = 0, W = 0;
    int refs = 0;
    float* stats = nullptr;   // GroupNorm partials written by the producing convolution (sisic_conv_args.stats_out)
    int slots = 0;
    const void* fin_norm = nullptr;   // the GroupNorm module whose (scale, shift) the producing convolution has already left in
    float* fin_scale = nullptr;       // fin_scale / fin_shift (sisic_conv_args.fin_*): its finalisation launch is skipped
    float* fin_shift = nullptr;
};

struct PoolBlock {
    float* p;
    size_t bytes;
    bool free_;
};

// One recorded operation of a training-mode forward pass (train.cpp walks the tape backwards).
struct TapeOp {
    enum Kind { CONV = 0, ATTN = 1 };
    int kind = CONV;
    // ---- CONV: out = conv(act(cat(in0, in1))) + bias + tproj[:, temb_off:] + residual
    ConvW w;                          // by value: the fused q/k/v projection has no ConvW of its own
    const AttnW* qkv_of = nullptr;    // set for the fused q/k/v projection (its gradient is split three ways)
    Buf* in0 = nullptr;               // nullptr: the network input (no data gradient is needed)
    Buf* in1 = nullptr;
    const float* in0_ptr = nullptr;
    const float* in1_ptr = nullptr;
    int c0 = 0, c1 = 0, H = 0, W = 0, stride = 1, ups = 0;
    const NormW* norm = nullptr;      // GroupNorm prologue (its gamma/beta gradients), or nullptr
    float* gn_scale = nullptr;        // saved [B, c0+c1] of this op
    float* gn_shift = nullptr;
    float* gn_mr = nullptr;           // saved (mean, rstd) [B, groups, 2]
    bool silu = false;
    int temb_off = -1;                // >= 0: the time-embedding projection columns added per (sample, channel)
    Buf* residual = nullptr;
    Buf* out = nullptr;               // nullptr: the network output
    float* out_ptr = nullptr;
    // ---- ATTN: o = softmax(q k^T / sqrt(d)) v per head
    Buf* qkv = nullptr;
    Buf* o = nullptr;
    int C = 0, N = 0;
};

// Everything a training run keeps between calls (allocated by sisic_unet_train_begin).
struct TrainState {
    float* grad = nullptr;            // gradient arena, same layout as sisic_unet::raw
    float* adam_m = nullptr;
    float* adam_v = nullptr;
    int64_t step = 0;                 // optimizer steps taken
    // tape of the last training-mode forward
    std::vector<TapeOp> tape;
    std::vector<std::unique_ptr<Buf>> bufs;
    std::vector<float*> grads_of_bufs;            // pool blocks holding activation gradients (returned after backward)
    std::map<const Buf*, float*> buf_grad;
    int B = 0, H = 0, W = 0;
    bool has_tape = false;
    // saved time-embedding intermediates of the tape [B, .]
    float* emb = nullptr;             // sinusoid [B, 2*n_freqs]
    float* h1 = nullptr;              // linear_1 output before SiLU [B, hidden]
    float* t2 = nullptr;              // linear_2 output before SiLU [B, hidden]
    size_t emb_cap = 0, h1_cap = 0, t2_cap = 0;
    float* dtproj = nullptr;          // [B, tproj_R] gradient of the fused time_emb_proj outputs
    size_t dtproj_cap = 0;
    float* garena = nullptr;          // activation gradients of one backward pass: one block, zero-filled once
    size_t garena_cap = 0, garena_used = 0;
    float* wgrad_part = nullptr;      // K-split partial weight gradients
    size_t wgrad_part_cap = 0;
    float* scratch = nullptr;         // data-gradient scratch [B, Cin, H, W] of the widest layer
    size_t scratch_cap = 0;
    float* small = nullptr;           // per-(sample, channel) sums and similar
    size_t small_cap = 0;
    float* loss_dev = nullptr;        // [2]: loss, scratch
    int* flag_dev = nullptr;          // non-finite gradient flag
    float* mse_part = nullptr;
    // every re-layout of every weight as three batched launches (repack.hip); rebuilt when the derived buffers move
    void* repack_dev[3] = {nullptr, nullptr, nullptr};     // PackJob tables on the device, one per phase
    int repack_jobs[3] = {0, 0, 0}, repack_blocks[3] = {0, 0, 0};
    void* scatter_dev = nullptr;          // PackJob table: the fused time-embedding projection's gradient rows -> the 22 blocks' own tensors
    int scatter_jobs = 0, scatter_blocks = 0;
    const float* scatter_src = nullptr;   // the scratch address the table was built for
    bool repack_ready = false;
};

}  // namespace sisic

struct sisic_unet {
    using ConvW = sisic::ConvW; using NormW = sisic::NormW; using ResnetW = sisic::ResnetW; using AttnW = sisic::AttnW;
    using PoolBlock = sisic::PoolBlock;
    sisic_ctx* ctx = nullptr;
    sisic_unet_config cfg{};
    std::vector<float> freqs;

    // expected state dict
    std::vector<std::string> names;
    std::vector<int64_t> numels;
    std::vector<size_t> offsets;       // float offset of each raw tensor in `raw`
    std::map<std::string, int> index;
    float* raw = nullptr;              // device arena with the raw tensors
    size_t raw_floats = 0;
    std::vector<float*> owned;         // derived device buffers (packed weights, ...)
    bool loaded = false;

    // architecture
    ConvW conv_in, conv_out;
    NormW norm_out;
    int temb_w1 = -1, temb_b1 = -1, temb_w2 = -1, temb_b2 = -1;
    float* d_freqs = nullptr;
    float* w1t = nullptr;  // [2*n_freqs][hidden]
    float* w2t = nullptr;  // [hidden][hidden]
    float* tproj_wt = nullptr;   // [hidden][tproj_R]
    float* tproj_b = nullptr;    // [tproj_R]
    int hidden = 0, tproj_R = 0;
    std::vector<std::vector<ResnetW>> down_res, up_res;
    std::vector<std::vector<AttnW>> down_attn, up_attn;
    std::vector<ConvW> downsamplers, upsamplers;   // k==0 when absent
    ResnetW mid_res[2];
    AttnW mid_attn;
    int max_c = 0;

    // workspace
    std::vector<PoolBlock> pool;
    int ws_B = 0, ws_H = 0, ws_W = 0;
    float* t_vals = nullptr;     // [B] or [T]
    float* temb_act = nullptr;   // [B or T, hidden]
    float* tproj = nullptr;      // [B or T, tproj_R]
    float* gn_scale = nullptr;   // [B, max_c]
    float* gn_shift = nullptr;
    float* gn_scale2 = nullptr;  // a second pair: a convolution that finalizes the GroupNorm of its own output (sisic_conv_args.fin_*)
    float* gn_shift2 = nullptr;  //   writes the pair its own prologue is NOT reading (later workgroups of the launch still read that one)
    size_t t_vals_cap = 0, temb_act_cap = 0, tproj_cap = 0, gn_scale_cap = 0, gn_shift_cap = 0, gn_scale2_cap = 0, gn_shift2_cap = 0;
    static constexpr int STAGE_SLOTS = 4;
    float* stage_host = nullptr; // pinned upload ring
    size_t stage_cap = 0;
    uint64_t stage_next = 0;
    hipEvent_t stage_ev[STAGE_SLOTS] = {};
    bool stage_used[STAGE_SLOTS] = {};
    bool latency_mode = false;      // sisic_unet_set_latency_mode: per-layer kernels chosen so that ONE image fills the chip
    bool use_winograd = true;
    bool fuse_gn = true;            // GroupNorm statistics from convolution epilogues where the kernel offers them
                                    // (SISIC_FUSED_GN=0 in the environment: always the stand-alone statistics pass)
    // SISIC_WINOGRAD=0 in the environment keeps every 3x3 on the direct kernel
    float* eps_buf = nullptr;    // sampling loop scratch [B,C,H,W]
    size_t eps_floats = 0;
    std::unique_ptr<sisic::TrainState> train;     // present after sisic_unet_train_begin

    // graph-replayed sampling loop (sisic_sample): one captured step, replayed T-1 times
    int graph_mode = -1;                 // -1: follow latency_mode (SISIC_GRAPH in the environment: 0 / 1 forces)
    hipGraph_t loop_graph = nullptr;
    hipGraphExec_t loop_exec = nullptr;
    struct LoopKey {
        int B = 0, H = 0, W = 0; float clip = 0; hipStream_t s = nullptr; bool latency = false; uint64_t gen = 0;
        const void* ptrs[5] = {};        // tproj, eps_buf, x_work, loop_tables, tproj_cur at capture time: each can be re-allocated
    } loop_key;
    bool loop_valid = false;
    int64_t loop_builds = 0;             // captures + instantiations so far (sisic_unet_graph_builds)
    hipStream_t loop_stream = nullptr;   // used when the caller's stream is the legacy default stream (not capturable)
    float* x_work = nullptr;             // the loop's own latent buffer: every address inside the graph is library-owned
    size_t x_work_cap = 0;
    float* loop_tables = nullptr;        // [state 4 floats][coef 5*T][zrow T] device
    size_t loop_tables_cap = 0;
    float* tproj_cur = nullptr;
    size_t tproj_cur_cap = 0;

    int add(const std::string& name, int64_t numel) {
        index[name] = (int)names.size();
        names.push_back(name);
        numels.push_back(numel);
        return (int)names.size() - 1;
    }
    const float* rawp(int idx) const { return raw + offsets[idx]; }
};


namespace sisic {

// unet.cpp internals used by train.cpp
int unet_pool_get(sisic_unet* u, size_t floats, float** out);
void unet_pool_put(sisic_unet* u, float* p);
// hand the recorded training forward's blocks back to the pool and forget it (backward then answers SISIC_ESTATE)
void unet_release_tape(sisic_unet* u);
// floats of TrainState::small: [B, Cout <= 3 max_c] plane sums, then [2][B][Cin <= max_c] GroupNorm sums or [3C] column sums
inline size_t train_small_floats(const sisic_unet* u, int B) {
    return (size_t)B * 5 * (size_t)u->max_c + 3 * (size_t)u->max_c + 2 * (size_t)B;
}
int unet_grow(float** p, size_t* have, size_t want);
int unet_check_shape(sisic_unet* u, int B, int H, int W);
int unet_ensure_rows(sisic_unet* u, size_t t_rows, size_t gn_rows);
int unet_stage_upload(sisic_unet* u, const float* src, size_t n, float* dst, hipStream_t s);
// (re)derive every packed / transposed weight buffer from the raw arena (after load, and after each optimizer step)
int unet_prepare_all(sisic_unet* u, hipStream_t s);
// forward pass; with `tape` set nothing is released and every operation is recorded (training mode)
int unet_run_forward(sisic_unet* u, const float* sample, const float* tproj, int tproj_stride, float* out, int B, int H,
                     int W, hipStream_t s, TrainState* tape);

// every convolution / attention block / residual block of the model, in state-dict order
inline std::vector<ResnetW*> unet_resnets(sisic_unet* u) {
    std::vector<ResnetW*> v;
    for (auto& blk : u->down_res) for (auto& r : blk) v.push_back(&r);
    for (auto& r : u->mid_res) v.push_back(&r);
    for (auto& blk : u->up_res) for (auto& r : blk) v.push_back(&r);
    return v;
}
inline std::vector<AttnW*> unet_attns(sisic_unet* u) {
    std::vector<AttnW*> v;
    for (auto& blk : u->down_attn) for (auto& a : blk) v.push_back(&a);
    v.push_back(&u->mid_attn);
    for (auto& blk : u->up_attn) for (auto& a : blk) v.push_back(&a);
    return v;
}
inline std::vector<ConvW*> unet_convs(sisic_unet* u) {
    std::vector<ConvW*> v{&u->conv_in, &u->conv_out};
    for (auto* r : unet_resnets(u)) {
        v.push_back(&r->conv1); v.push_back(&r->conv2);
        if (r->shortcut.k) v.push_back(&r->shortcut);
    }
    for (auto* a : unet_attns(u)) v.push_back(&a->out);
    for (auto& c : u->downsamplers) if (c.k) v.push_back(&c);
    for (auto& c : u->upsamplers) if (c.k) v.push_back(&c);
    return v;
}

}  // namespace sisic
