// unet.cpp -- the UNet2DModel executor and the reverse-diffusion loop.
//
// Walks the architecture the reference configures at core/generator/model_manager.py:173-194
//  and issues the fused HIP kernels:
//     ResnetBlock2D  = gn_stats -> conv3x3[GN+SiLU prologue, +bias +time-embedding]
//                      -> gn_stats -> conv3x3[GN+SiLU prologue, +bias +shortcut/residual]
//                      (conv1x1 shortcut on the raw, possibly concatenated input when Cin != Cout)
//     Attention      = gn_stats -> conv1x1[GN prologue] (q,k,v in one launch) -> attention core
//                      -> conv1x1[+bias +residual]
//     Downsample2D   = conv3x3 stride 2;   Upsample2D = conv3x3 reading through a nearest-2x map
// torch.cat of the skip connections never materialises: consumers read two source tensors.
// sisic_sample() runs the loop of core/generator/image_generator.py:395-403 without returning
// to the host between steps.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>

#include "unet_internal.h"

using namespace sisic;

namespace {

// ------------------------------------------------------------------ architecture description
void add_conv(sisic_unet* u, ConvW& c, const std::string& name, int cout, int cin, int k) {
    c.cout = cout; c.cin = cin; c.k = k;
    c.w_idx = u->add(name + ".weight", (int64_t)cout * cin * k * k);
    c.b_idx = u->add(name + ".bias", cout);
}
void add_norm(sisic_unet* u, NormW& n, const std::string& name, int c) {
    n.c = c;
    n.w_idx = u->add(name + ".weight", c);
    n.b_idx = u->add(name + ".bias", c);
}
void add_resnet(sisic_unet* u, ResnetW& r, const std::string& name, int cin, int cout) {
    r.cin = cin; r.cout = cout;
    add_norm(u, r.norm1, name + ".norm1", cin);
    add_conv(u, r.conv1, name + ".conv1", cout, cin, 3);
    r.temb_w_idx = u->add(name + ".time_emb_proj.weight", (int64_t)cout * u->hidden);
    r.temb_b_idx = u->add(name + ".time_emb_proj.bias", cout);
    r.temb_off = u->tproj_R;
    u->tproj_R += cout;
    add_norm(u, r.norm2, name + ".norm2", cout);
    add_conv(u, r.conv2, name + ".conv2", cout, cout, 3);
    if (cin != cout) add_conv(u, r.shortcut, name + ".conv_shortcut", cout, cin, 1);
    u->max_c = std::max(u->max_c, std::max(cin, cout));
}
void add_attn(sisic_unet* u, AttnW& a, const std::string& name, int c) {
    a.c = c;
    add_norm(u, a.norm, name + ".group_norm", c);
    a.q_w = u->add(name + ".to_q.weight", (int64_t)c * c); a.q_b = u->add(name + ".to_q.bias", c);
    a.k_w = u->add(name + ".to_k.weight", (int64_t)c * c); a.k_b = u->add(name + ".to_k.bias", c);
    a.v_w = u->add(name + ".to_v.weight", (int64_t)c * c); a.v_b = u->add(name + ".to_v.bias", c);
    add_conv(u, a.out, name + ".to_out.0", c, c, 1);
    u->max_c = std::max(u->max_c, c);
}

int describe(sisic_unet* u) {
    const sisic_unet_config& cfg = u->cfg;
    const int n = cfg.n_blocks;
    const int* boc = cfg.block_out_channels;
    u->hidden = 4 * boc[0];
    add_conv(u, u->conv_in, "conv_in", boc[0], cfg.in_channels, 3);
    u->temb_w1 = u->add("time_embedding.linear_1.weight", (int64_t)u->hidden * boc[0]);
    u->temb_b1 = u->add("time_embedding.linear_1.bias", u->hidden);
    u->temb_w2 = u->add("time_embedding.linear_2.weight", (int64_t)u->hidden * u->hidden);
    u->temb_b2 = u->add("time_embedding.linear_2.bias", u->hidden);

    u->down_res.resize(n); u->down_attn.resize(n); u->downsamplers.resize(n);
    int out_ch = boc[0];
    for (int i = 0; i < n; ++i) {
        const int in_ch = out_ch;
        out_ch = boc[i];
        u->down_res[i].resize(cfg.layers_per_block);
        if (cfg.down_attn[i]) u->down_attn[i].resize(cfg.layers_per_block);
        for (int j = 0; j < cfg.layers_per_block; ++j) {
            const std::string base = "down_blocks." + std::to_string(i);
            add_resnet(u, u->down_res[i][j], base + ".resnets." + std::to_string(j), j == 0 ? in_ch : out_ch, out_ch);
            if (cfg.down_attn[i]) add_attn(u, u->down_attn[i][j], base + ".attentions." + std::to_string(j), out_ch);
        }
        if (i != n - 1)
            add_conv(u, u->downsamplers[i], "down_blocks." + std::to_string(i) + ".downsamplers.0.conv", out_ch, out_ch, 3);
    }
    const int mid = boc[n - 1];
    add_resnet(u, u->mid_res[0], "mid_block.resnets.0", mid, mid);
    add_attn(u, u->mid_attn, "mid_block.attentions.0", mid);
    add_resnet(u, u->mid_res[1], "mid_block.resnets.1", mid, mid);

    u->up_res.resize(n); u->up_attn.resize(n); u->upsamplers.resize(n);
    out_ch = boc[n - 1];
    for (int i = 0; i < n; ++i) {
        const int prev_out = out_ch;
        out_ch = boc[n - 1 - i];
        const int in_ch = boc[n - 1 - std::min(i + 1, n - 1)];
        const int layers = cfg.layers_per_block + 1;
        u->up_res[i].resize(layers);
        if (cfg.up_attn[i]) u->up_attn[i].resize(layers);
        for (int j = 0; j < layers; ++j) {
            const int skip_ch = (j == layers - 1) ? in_ch : out_ch;
            const int res_in = (j == 0) ? prev_out : out_ch;
            const std::string base = "up_blocks." + std::to_string(i);
            add_resnet(u, u->up_res[i][j], base + ".resnets." + std::to_string(j), res_in + skip_ch, out_ch);
            if (cfg.up_attn[i]) add_attn(u, u->up_attn[i][j], base + ".attentions." + std::to_string(j), out_ch);
        }
        if (i != n - 1)
            add_conv(u, u->upsamplers[i], "up_blocks." + std::to_string(i) + ".upsamplers.0.conv", out_ch, out_ch, 3);
    }
    add_norm(u, u->norm_out, "conv_norm_out", boc[0]);
    add_conv(u, u->conv_out, "conv_out", cfg.out_channels, boc[0], 3);

    u->offsets.resize(u->names.size());
    size_t off = 0;
    for (size_t i = 0; i < u->names.size(); ++i) {
        u->offsets[i] = off;
        off += ((size_t)u->numels[i] + 3) / 4 * 4;   // keep every tensor 16-byte aligned
    }
    u->raw_floats = off;
    return SISIC_OK;
}

// ------------------------------------------------------------------ derived weights
int dev_alloc(sisic_unet* u, size_t floats, float** out) {
    if (*out) return SISIC_OK;          // already there: prepare_* runs again after every optimizer step
    void* p = nullptr;
    SISIC_HIP(hipMalloc(&p, std::max<size_t>(floats, 4) * sizeof(float)));
    u->owned.push_back(static_cast<float*>(p));
    *out = static_cast<float*>(p);
    return SISIC_OK;
}

int prepare_conv(sisic_unet* u, ConvW& c, hipStream_t s) {
    if (c.k == 0) return SISIC_OK;
    SISIC_TRY(dev_alloc(u, (size_t)sisic_conv_packed_numel(c.cout, c.cin, c.k), &c.packed));
    SISIC_TRY(launch_conv_pack(u->ctx, u->rawp(c.w_idx), c.cout, c.cin, c.k, c.packed, s));
    c.bias = u->rawp(c.b_idx);
    if (c.k == 3 && !c.strided && c.cout > 4) {
        SISIC_TRY(dev_alloc(u, (size_t)winograd_packed_numel(c.cout, c.cin), &c.wino));
        SISIC_TRY(launch_winograd_pack(u->ctx, u->rawp(c.w_idx), c.cout, c.cin, c.wino, s));
    }
    return SISIC_OK;
}
void prepare_norm(sisic_unet* u, NormW& n) {
    n.gamma = u->rawp(n.w_idx);
    n.beta = u->rawp(n.b_idx);
}
int prepare_resnet(sisic_unet* u, ResnetW& r, hipStream_t s) {
    prepare_norm(u, r.norm1);
    prepare_norm(u, r.norm2);
    SISIC_TRY(prepare_conv(u, r.conv1, s));
    SISIC_TRY(prepare_conv(u, r.conv2, s));
    SISIC_TRY(prepare_conv(u, r.shortcut, s));
    // time_emb_proj.weight [cout, hidden] -> columns [temb_off, temb_off+cout) of tproj_wt [hidden][R]
    SISIC_TRY(launch_transpose2d(u->ctx, u->rawp(r.temb_w_idx), r.cout, u->hidden, u->tproj_wt, u->tproj_R, r.temb_off, s));
    SISIC_HIP(hipMemcpyAsync(u->tproj_b + r.temb_off, u->rawp(r.temb_b_idx), (size_t)r.cout * sizeof(float), hipMemcpyDeviceToDevice, s));
    return SISIC_OK;
}
int prepare_attn(sisic_unet* u, AttnW& a, hipStream_t s) {
    prepare_norm(u, a.norm);
    const int c = a.c;
    // q,k,v as one [3C, C] 1x1 convolution
    SISIC_TRY(dev_alloc(u, (size_t)3 * c * c, &a.qkv_cat));
    float* cat = a.qkv_cat;
    const size_t wbytes = (size_t)c * c * sizeof(float);
    SISIC_HIP(hipMemcpyAsync(cat, u->rawp(a.q_w), wbytes, hipMemcpyDeviceToDevice, s));
    SISIC_HIP(hipMemcpyAsync(cat + (size_t)c * c, u->rawp(a.k_w), wbytes, hipMemcpyDeviceToDevice, s));
    SISIC_HIP(hipMemcpyAsync(cat + (size_t)2 * c * c, u->rawp(a.v_w), wbytes, hipMemcpyDeviceToDevice, s));
    SISIC_TRY(dev_alloc(u, (size_t)sisic_conv_packed_numel(3 * c, c, 1), &a.qkv_packed));
    SISIC_TRY(launch_conv_pack(u->ctx, cat, 3 * c, c, 1, a.qkv_packed, s));
    SISIC_TRY(dev_alloc(u, (size_t)3 * c, &a.qkv_bias));
    SISIC_HIP(hipMemcpyAsync(a.qkv_bias, u->rawp(a.q_b), c * sizeof(float), hipMemcpyDeviceToDevice, s));
    SISIC_HIP(hipMemcpyAsync(a.qkv_bias + c, u->rawp(a.k_b), c * sizeof(float), hipMemcpyDeviceToDevice, s));
    SISIC_HIP(hipMemcpyAsync(a.qkv_bias + 2 * c, u->rawp(a.v_b), c * sizeof(float), hipMemcpyDeviceToDevice, s));
    SISIC_TRY(prepare_conv(u, a.out, s));
    return SISIC_OK;
}

int prepare_all(sisic_unet* u, hipStream_t s) {
    const int nin = 2 * u->cfg.n_freqs;
    SISIC_TRY(dev_alloc(u, u->cfg.n_freqs, &u->d_freqs));
    SISIC_HIP(hipMemcpyAsync(u->d_freqs, u->freqs.data(), u->freqs.size() * sizeof(float), hipMemcpyHostToDevice, s));
    SISIC_TRY(dev_alloc(u, (size_t)nin * u->hidden, &u->w1t));
    SISIC_TRY(dev_alloc(u, (size_t)u->hidden * u->hidden, &u->w2t));
    SISIC_TRY(dev_alloc(u, (size_t)u->hidden * u->tproj_R, &u->tproj_wt));
    SISIC_TRY(dev_alloc(u, (size_t)u->tproj_R, &u->tproj_b));
    SISIC_TRY(launch_transpose2d(u->ctx, u->rawp(u->temb_w1), u->hidden, nin, u->w1t, u->hidden, 0, s));
    SISIC_TRY(launch_transpose2d(u->ctx, u->rawp(u->temb_w2), u->hidden, u->hidden, u->w2t, u->hidden, 0, s));
    SISIC_TRY(prepare_conv(u, u->conv_in, s));
    SISIC_TRY(prepare_conv(u, u->conv_out, s));
    prepare_norm(u, u->norm_out);
    for (auto& blk : u->down_res) for (auto& r : blk) SISIC_TRY(prepare_resnet(u, r, s));
    for (auto& blk : u->up_res) for (auto& r : blk) SISIC_TRY(prepare_resnet(u, r, s));
    for (auto& r : u->mid_res) SISIC_TRY(prepare_resnet(u, r, s));
    for (auto& blk : u->down_attn) for (auto& a : blk) SISIC_TRY(prepare_attn(u, a, s));
    for (auto& blk : u->up_attn) for (auto& a : blk) SISIC_TRY(prepare_attn(u, a, s));
    SISIC_TRY(prepare_attn(u, u->mid_attn, s));
    for (auto& c : u->downsamplers) {
        c.strided = true;
        SISIC_TRY(prepare_conv(u, c, s));
    }
    for (auto& c : u->upsamplers) SISIC_TRY(prepare_conv(u, c, s));
    return SISIC_OK;
}

// forget every derived buffer (they were freed): the next prepare_all allocates afresh
void reset_conv(ConvW& c) { c.packed = c.wino = c.raw_t = c.packed_t = c.wino_t = nullptr; }
void reset_attn(AttnW& a) {
    a.qkv_cat = a.qkv_packed = a.qkv_bias = a.qkv_raw_t = a.qkv_packed_t = nullptr;
    reset_conv(a.out);
}
void reset_resnet(ResnetW& r) { reset_conv(r.conv1); reset_conv(r.conv2); reset_conv(r.shortcut); }
void reset_derived(sisic_unet* u) {
    if (u->train) u->train->repack_ready = false;       // the job tables of repack.hip name the old buffers
    u->d_freqs = u->w1t = u->w2t = u->tproj_wt = u->tproj_b = nullptr;
    reset_conv(u->conv_in); reset_conv(u->conv_out);
    for (auto& blk : u->down_res) for (auto& r : blk) reset_resnet(r);
    for (auto& blk : u->up_res) for (auto& r : blk) reset_resnet(r);
    for (auto& r : u->mid_res) reset_resnet(r);
    for (auto& blk : u->down_attn) for (auto& a : blk) reset_attn(a);
    for (auto& blk : u->up_attn) for (auto& a : blk) reset_attn(a);
    reset_attn(u->mid_attn);
    for (auto& c : u->downsamplers) reset_conv(c);
    for (auto& c : u->upsamplers) reset_conv(c);
}

// ------------------------------------------------------------------ workspace
void pool_put(sisic_unet* u, float* p);

void loop_graph_drop(sisic_unet* u) {
    if (u->loop_exec) (void)hipGraphExecDestroy(u->loop_exec);
    if (u->loop_graph) (void)hipGraphDestroy(u->loop_graph);
    u->loop_exec = nullptr; u->loop_graph = nullptr; u->loop_valid = false;
}

// A recorded training forward (the tape) owns pool blocks.  Whatever frees the pool, or lets a captured step write into
// blocks the tape took from the free list after the capture, makes the tape unusable: forget it, so that
// sisic_unet_backward answers SISIC_ESTATE instead of reading freed or overwritten activations.
void tape_drop(sisic_unet* u, bool return_blocks) {
    TrainState* tr = u->train.get();
    if (!tr) return;
    if (return_blocks) {
        for (auto& b : tr->bufs) {
            if (b->p) pool_put(u, b->p);
            if (b->stats) pool_put(u, b->stats);
        }
        for (float* p : tr->grads_of_bufs) pool_put(u, p);
    }
    tr->bufs.clear();
    tr->grads_of_bufs.clear();
    tr->buf_grad.clear();
    tr->tape.clear();
    tr->has_tape = false;
}

void pool_release_all(sisic_unet* u) {
    loop_graph_drop(u);                 // a captured step holds addresses of pool blocks
    tape_drop(u, false);                // ... and so does a recorded training forward
    for (auto& b : u->pool) (void)hipFree(b.p);
    u->pool.clear();
}

int pool_get(sisic_unet* u, size_t floats, float** out) {
    const size_t bytes = floats * sizeof(float);
    for (auto& b : u->pool) {
        if (b.free_ && b.bytes == bytes) {
            b.free_ = false;
            *out = b.p;
            return SISIC_OK;
        }
    }
    void* p = nullptr;
    SISIC_HIP(hipMalloc(&p, bytes));
    u->pool.push_back({static_cast<float*>(p), bytes, false});
    *out = static_cast<float*>(p);
    return SISIC_OK;
}

void pool_put(sisic_unet* u, float* p) {
    for (auto& b : u->pool)
        if (b.p == p) { b.free_ = true; return; }
}

int grow(float** p, size_t* have, size_t want) {
    if (*have >= want) return SISIC_OK;
    if (*p) SISIC_HIP(hipFree(*p));   // hipFree waits for the device, nothing can still read the old block
    *p = nullptr; *have = 0;
    void* q = nullptr;
    SISIC_HIP(hipMalloc(&q, want * sizeof(float)));
    *p = static_cast<float*>(q);
    *have = want;
    return SISIC_OK;
}

int ensure_rows(sisic_unet* u, size_t t_rows, size_t gn_rows) {
    SISIC_TRY(grow(&u->t_vals, &u->t_vals_cap, t_rows));
    SISIC_TRY(grow(&u->temb_act, &u->temb_act_cap, t_rows * u->hidden));
    SISIC_TRY(grow(&u->tproj, &u->tproj_cap, t_rows * u->tproj_R));
    SISIC_TRY(grow(&u->gn_scale, &u->gn_scale_cap, gn_rows * u->max_c));
    SISIC_TRY(grow(&u->gn_shift, &u->gn_shift_cap, gn_rows * u->max_c));
    SISIC_TRY(grow(&u->gn_scale2, &u->gn_scale2_cap, gn_rows * u->max_c));
    SISIC_TRY(grow(&u->gn_shift2, &u->gn_shift2_cap, gn_rows * u->max_c));
    return SISIC_OK;
}

// Host -> device upload of a few floats, ordered on the caller's stream.  The source is copied into
// one of a small ring of pinned slots; a slot is reused only after the copy that read it has finished.
int stage_upload(sisic_unet* u, const float* src, size_t n, float* dst, hipStream_t s) {
    if (u->stage_cap < n) {
        SISIC_HIP(hipDeviceSynchronize());
        if (u->stage_host) SISIC_HIP(hipHostFree(u->stage_host));
        u->stage_host = nullptr;
        const size_t cap = std::max<size_t>(n, 1024);
        void* p = nullptr;
        SISIC_HIP(hipHostMalloc(&p, cap * sisic_unet::STAGE_SLOTS * sizeof(float), hipHostMallocDefault));
        u->stage_host = static_cast<float*>(p);
        u->stage_cap = cap;
        for (int i = 0; i < sisic_unet::STAGE_SLOTS; ++i) u->stage_used[i] = false;
    }
    const int slot = (int)(u->stage_next++ % sisic_unet::STAGE_SLOTS);
    if (!u->stage_ev[slot]) SISIC_HIP(hipEventCreateWithFlags(&u->stage_ev[slot], hipEventDisableTiming));
    if (u->stage_used[slot]) SISIC_HIP(hipEventSynchronize(u->stage_ev[slot]));
    float* h = u->stage_host + (size_t)slot * u->stage_cap;
    std::memcpy(h, src, n * sizeof(float));
    SISIC_HIP(hipMemcpyAsync(dst, h, n * sizeof(float), hipMemcpyHostToDevice, s));
    SISIC_HIP(hipEventRecord(u->stage_ev[slot], s));
    u->stage_used[slot] = true;
    return SISIC_OK;
}

// ------------------------------------------------------------------ forward
struct Fwd {
    sisic_unet* u;
    hipStream_t s;
    int B;
    const float* tproj;   // [B or 1, tproj_R]
    int tproj_stride;     // tproj_R, or 0 when one row serves every sample
    std::vector<std::unique_ptr<Buf>> bufs;
    TrainState* tr = nullptr;     // training mode: record every operation, keep every buffer, save each GroupNorm's statistics
    float* gsc = nullptr;         // scale / shift the next convolution's prologue reads (the shared pair, or this op's own)
    float* gsh = nullptr;
    float* gmr = nullptr;         // training mode: (mean, rstd) of the same GroupNorm
    const NormW* gnorm = nullptr;

    Buf* make(int C, int H, int W, int* rc) {
        auto b = std::make_unique<Buf>();
        b->C = C; b->H = H; b->W = W; b->refs = 1;
        *rc = pool_get(u, (size_t)B * C * H * W, &b->p);
        bufs.push_back(std::move(b));
        return bufs.back().get();
    }
    void release(Buf* b) {
        if (tr) return;           // the backward pass reads every activation
        if (b && --b->refs == 0 && b->p) {
            pool_put(u, b->p);
            b->p = nullptr;
            if (b->stats) { pool_put(u, b->stats); b->stats = nullptr; }
        }
    }

    int gn(const Buf* x, const Buf* skip, const NormW& n) {
        gsc = u->gn_scale; gsh = u->gn_shift; gmr = nullptr; gnorm = &n;
        if (!tr && !skip && x->fin_norm == static_cast<const void*>(&n)) {     // finalized by its producer (conv below)
            gsc = x->fin_scale; gsh = x->fin_shift;
            return SISIC_OK;
        }
        if (tr) {                 // this GroupNorm's own scale / shift / (mean, rstd): the backward pass needs them
            const int C = x->C + (skip ? skip->C : 0);
            SISIC_TRY(pool_get(u, (size_t)B * C, &gsc));
            SISIC_TRY(pool_get(u, (size_t)B * C, &gsh));
            SISIC_TRY(pool_get(u, (size_t)B * u->cfg.norm_groups * 2, &gmr));
            tr->grads_of_bufs.push_back(gsc); tr->grads_of_bufs.push_back(gsh); tr->grads_of_bufs.push_back(gmr);
        }
        if (x->stats && (!skip || skip->stats))   // every producer left partials: no pass over the tensors
            return launch_gn_finalize(u->ctx, x->stats, x->C, x->slots, skip ? skip->stats : nullptr, skip ? skip->C : 0,
                                      skip ? skip->slots : 0, B, x->H * x->W, u->cfg.norm_groups, u->cfg.norm_eps,
                                      n.gamma, n.beta, gsc, gsh, s, gmr);
        return launch_gn_stats(u->ctx, x->p, x->C, skip ? skip->p : nullptr, skip ? skip->C : 0, B, x->H * x->W,
                               u->cfg.norm_groups, u->cfg.norm_eps, n.gamma, n.beta, gsc, gsh, s, gmr);
    }

    // xb / skipb / resb / outb: the buffers behind in0 / in1 / residual / out (training tape; nullptr = network input / output)
    int conv(const ConvW& c, const float* in0, int c0, const float* in1, int c1, int H, int W, int stride, int ups,
             bool gn_prologue, bool silu, const float* chan_bias, const float* residual, float* out,
             Buf* normed_later = nullptr, Buf* xb = nullptr, Buf* skipb = nullptr, Buf* resb = nullptr, Buf* outb = nullptr,
             const AttnW* qkv_of = nullptr, const NormW* next_norm = nullptr) {
        sisic_conv_args a{};
        a.in0 = in0; a.c0 = c0; a.in1 = in1; a.c1 = c1;
        a.B = B; a.Hin = H; a.Win = W; a.upsample = ups; a.ksize = c.k; a.stride = stride;
        a.w_packed = c.packed; a.bias = c.bias; a.Cout = c.cout;
        a.w_winograd = (stride == 1 && u->use_winograd) ? c.wino : nullptr;
        if (gn_prologue) { a.gn_scale = gsc; a.gn_shift = gsh; a.gn_silu = silu ? 1 : 0; }
        a.chan_bias = chan_bias; a.chan_bias_stride = tproj_stride;
        a.residual = residual; a.out = out;
        if (u->latency_mode) {
            // single-image latency (the reference's own call pattern, image_generator.py:379: batch 1): the Winograd
            // convolutions split their input channels over workgroups (tile_cfg 78 / 79) and the 1x1 convolutions take the
            // 64-pixel tiles, so that one image offers a few hundred workgroups per layer.  Chosen from the layer shape only:
            // inside this mode an image's bits are again independent of the batch.
            // (the 64-channel, two-workgroups-per-CU form for every Cout: at one image it is level with or ahead of the
            //  128-channel form on every layer, profiles/r02/ksplit_in_place_ab.txt rows 76 (128-channel) / 77 (64-channel))
            if (c.k == 3 && stride == 1 && !ups && a.w_winograd && c.cout > 4 && H >= 12 && W >= 12) a.tile_cfg = 79;
            else if (c.k == 1 && stride == 1) a.tile_cfg = 22;
            else if (c.k == 3 && stride == 2) a.tile_cfg = 18;      // 8x8-pixel tiles, two K groups of waves: 189 -> 105 us for the three downsamplers of one 128x128 image
        }
        if (normed_later && u->fuse_gn) {          // a GroupNorm reads this output: have the epilogue leave partials
            const int slots = conv_stats_slots(a);
            if (slots > 0) {
                SISIC_TRY(pool_get(u, (size_t)B * c.cout * slots * 4, &normed_later->stats));
                normed_later->slots = slots;
                a.stats_out = normed_later->stats;
            }
        }
        // next_norm: the GroupNorm that reads this output ALONE, when the caller knows it: where the launch can finalize it
        // (sisic_conv_finalizes: the 8x8 level's K-split forms) it leaves that layer's (scale, shift) in the shared pair -- which
        // this very convolution's prologue may still be reading: the finalisation runs in the reduction launch behind it
        if (normed_later && next_norm && u->fuse_gn && !tr) {
            a.fin_gamma = next_norm->gamma; a.fin_beta = next_norm->beta; a.fin_groups = u->cfg.norm_groups; a.fin_eps = u->cfg.norm_eps;
            // (the pair this launch's own prologue is NOT reading: a workgroup finalizes its image while others still read theirs)
            const bool first_pair_busy = gn_prologue && gsc == u->gn_scale;
            a.fin_scale = first_pair_busy ? u->gn_scale2 : u->gn_scale;
            a.fin_shift = first_pair_busy ? u->gn_shift2 : u->gn_shift;
            if (conv_finalizes(a)) { normed_later->fin_norm = next_norm; normed_later->fin_scale = a.fin_scale; normed_later->fin_shift = a.fin_shift; }
            else { a.fin_gamma = nullptr; a.fin_beta = nullptr; a.fin_scale = nullptr; a.fin_shift = nullptr; }
        }
        if (tr) {
            TapeOp op;
            op.kind = TapeOp::CONV;
            op.w = c; op.qkv_of = qkv_of;
            op.in0 = xb; op.in1 = skipb; op.in0_ptr = in0; op.in1_ptr = in1;
            op.c0 = c0; op.c1 = c1; op.H = H; op.W = W; op.stride = stride; op.ups = ups;
            if (gn_prologue) { op.norm = gnorm; op.gn_scale = gsc; op.gn_shift = gsh; op.gn_mr = gmr; op.silu = silu; }
            op.temb_off = chan_bias ? (int)(chan_bias - tproj) : -1;
            op.residual = resb; op.out = outb; op.out_ptr = out;
            tr->tape.push_back(op);
        }
        return launch_conv2d(u->ctx, a, s);
    }

    // out = ResnetBlock2D(cat(x, skip)); consumes nothing (caller releases inputs)
    // next_norm: the GroupNorm module that reads the block's output alone, when the caller knows it (conv(): finalized by conv2's
    // own launch where that is possible)
    int resnet(const ResnetW& r, const Buf* x, const Buf* skip, Buf** out, const NormW* next_norm = nullptr) {
        int rc = SISIC_OK;
        const int H = x->H, W = x->W;
        const int c1 = skip ? skip->C : 0;
        SISIC_REQUIRE(x->C + c1 == r.cin, "unet: resnet expects %d input channels, got %d", r.cin, x->C + c1);
        SISIC_TRY(gn(x, skip, r.norm1));
        Buf* h = make(r.cout, H, W, &rc); SISIC_TRY(rc);
        SISIC_TRY(conv(r.conv1, x->p, x->C, skip ? skip->p : nullptr, c1, H, W, 1, 0, true, true,
                       tproj + r.temb_off, nullptr, h->p, h, const_cast<Buf*>(x), const_cast<Buf*>(skip), nullptr, h, nullptr, &r.norm2));
        const float* residual = x->p;
        Buf* resb = const_cast<Buf*>(x);
        Buf* sc = nullptr;
        if (r.shortcut.k) {
            sc = make(r.cout, H, W, &rc); SISIC_TRY(rc);
            SISIC_TRY(conv(r.shortcut, x->p, x->C, skip ? skip->p : nullptr, c1, H, W, 1, 0, false, false, nullptr,
                           nullptr, sc->p, nullptr, const_cast<Buf*>(x), const_cast<Buf*>(skip), nullptr, sc));
            residual = sc->p;
            resb = sc;
        } else {
            SISIC_REQUIRE(c1 == 0 && x->C == r.cout, "unet: identity shortcut with mismatched channels");
        }
        SISIC_TRY(gn(h, nullptr, r.norm2));
        Buf* o = make(r.cout, H, W, &rc); SISIC_TRY(rc);
        SISIC_TRY(conv(r.conv2, h->p, r.cout, nullptr, 0, H, W, 1, 0, true, true, nullptr, residual, o->p, o, h, nullptr,
                       resb, o, nullptr, next_norm));
        release(h);
        release(sc);
        *out = o;
        return SISIC_OK;
    }

    int attention(const AttnW& a, const Buf* x, Buf** out) {
        int rc = SISIC_OK;
        const int H = x->H, W = x->W, N = H * W, C = a.c;
        SISIC_REQUIRE(x->C == C, "unet: attention channel mismatch");
        SISIC_TRY(gn(x, nullptr, a.norm));
        Buf* qkv = make(3 * C, H, W, &rc); SISIC_TRY(rc);
        ConvW cq; cq.cout = 3 * C; cq.cin = C; cq.k = 1; cq.packed = a.qkv_packed; cq.bias = a.qkv_bias;
        SISIC_TRY(conv(cq, x->p, C, nullptr, 0, H, W, 1, 0, true, false, nullptr, nullptr, qkv->p, nullptr,
                       const_cast<Buf*>(x), nullptr, nullptr, qkv, &a));
        Buf* o = make(C, H, W, &rc); SISIC_TRY(rc);
        SISIC_TRY(launch_attention(u->ctx, qkv->p, o->p, B, C, N, u->cfg.head_dim, s));
        if (tr) {
            TapeOp op;
            op.kind = TapeOp::ATTN;
            op.qkv = qkv; op.o = o; op.C = C; op.N = N;
            tr->tape.push_back(op);
        }
        release(qkv);
        Buf* y = make(C, H, W, &rc); SISIC_TRY(rc);
        SISIC_TRY(conv(a.out, o->p, C, nullptr, 0, H, W, 1, 0, false, false, nullptr, x->p, y->p, y, o, nullptr,
                       const_cast<Buf*>(x), y));
        release(o);
        *out = y;
        return SISIC_OK;
    }

    int run(const float* sample, float* out, int H, int W) {
        const sisic_unet_config& cfg = u->cfg;
        const int n = cfg.n_blocks;
        int rc = SISIC_OK;
        std::vector<Buf*> skips;

        Buf* x = make(u->conv_in.cout, H, W, &rc); SISIC_TRY(rc);
        SISIC_TRY(conv(u->conv_in, sample, cfg.in_channels, nullptr, 0, H, W, 1, 0, false, false, nullptr, nullptr, x->p, x,
                       nullptr, nullptr, nullptr, x));
        x->refs++;                 // held by `x` and by the skip stack
        skips.push_back(x);

        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < cfg.layers_per_block; ++j) {
                Buf* y = nullptr;
                // who normalises this block's output next (alone: the up path's concatenations have two producers)
                const NormW* next = cfg.down_attn[i] ? &u->down_attn[i][j].norm
                                    : (j + 1 < cfg.layers_per_block ? &u->down_res[i][j + 1].norm1 : (i == n - 1 ? &u->mid_res[0].norm1 : nullptr));
                SISIC_TRY(resnet(u->down_res[i][j], x, nullptr, &y, next));
                release(x);
                x = y;
                if (cfg.down_attn[i]) {
                    SISIC_TRY(attention(u->down_attn[i][j], x, &y));
                    release(x);
                    x = y;
                }
                x->refs++;
                skips.push_back(x);
            }
            if (i != n - 1) {
                const int Ho = (x->H + 2 - 3) / 2 + 1, Wo = (x->W + 2 - 3) / 2 + 1;
                Buf* y = make(u->downsamplers[i].cout, Ho, Wo, &rc); SISIC_TRY(rc);
                SISIC_TRY(conv(u->downsamplers[i], x->p, x->C, nullptr, 0, x->H, x->W, 2, 0, false, false, nullptr, nullptr, y->p, y,
                               x, nullptr, nullptr, y));
                release(x);
                x = y;
                x->refs++;
                skips.push_back(x);
            }
        }

        {
            Buf* y = nullptr;
            SISIC_TRY(resnet(u->mid_res[0], x, nullptr, &y, &u->mid_attn.norm)); release(x); x = y;
            SISIC_TRY(attention(u->mid_attn, x, &y)); release(x); x = y;
            SISIC_TRY(resnet(u->mid_res[1], x, nullptr, &y)); release(x); x = y;
        }

        for (int i = 0; i < n; ++i) {
            const int layers = cfg.layers_per_block + 1;
            for (int j = 0; j < layers; ++j) {
                SISIC_REQUIRE(!skips.empty(), "unet: skip stack underflow");
                Buf* skip = skips.back();
                skips.pop_back();
                SISIC_REQUIRE(skip->H == x->H && skip->W == x->W, "unet: skip resolution %dx%d vs %dx%d (H and W must be divisible by %d)",
                              skip->H, skip->W, x->H, x->W, 1 << (n - 1));
                Buf* y = nullptr;
                // (the block's output is normalised alone only when an attention block follows; the next ResNet block normalises
                //  it together with a skip tensor)
                SISIC_TRY(resnet(u->up_res[i][j], x, skip, &y, cfg.up_attn[i] ? &u->up_attn[i][j].norm : nullptr));
                release(x);
                release(skip);
                x = y;
                if (cfg.up_attn[i]) {
                    SISIC_TRY(attention(u->up_attn[i][j], x, &y));
                    release(x);
                    x = y;
                }
            }
            if (i != n - 1) {
                Buf* y = make(u->upsamplers[i].cout, 2 * x->H, 2 * x->W, &rc); SISIC_TRY(rc);
                SISIC_TRY(conv(u->upsamplers[i], x->p, x->C, nullptr, 0, x->H, x->W, 1, 1, false, false, nullptr, nullptr, y->p, y,
                               x, nullptr, nullptr, y));
                release(x);
                x = y;
            }
        }
        SISIC_REQUIRE(skips.empty(), "unet: skip stack not consumed");
        SISIC_REQUIRE(x->H == H && x->W == W, "unet: output resolution mismatch");
        SISIC_TRY(gn(x, nullptr, u->norm_out));
        SISIC_TRY(conv(u->conv_out, x->p, x->C, nullptr, 0, H, W, 1, 0, true, true, nullptr, nullptr, out, nullptr, x, nullptr,
                       nullptr, nullptr));
        release(x);
        return SISIC_OK;
    }
};

int check_shape(sisic_unet* u, int B, int H, int W) {
    SISIC_REQUIRE(u, "unet: null handle");
    if (!u->loaded) {
        set_error("unet: forward called before sisic_unet_load");
        return SISIC_ESTATE;
    }
    SISIC_HIP(hipSetDevice(u->ctx->device));
    const int div = 1 << (u->cfg.n_blocks - 1);
    SISIC_REQUIRE(B > 0 && H > 0 && W > 0 && H % div == 0 && W % div == 0,
                  "unet: sample %dx%dx%d unsupported (H and W must be positive multiples of %d)", B, H, W, div);
    if (u->ws_B != B || u->ws_H != H || u->ws_W != W) {
        SISIC_HIP(hipDeviceSynchronize());
        pool_release_all(u);
        u->ws_B = B; u->ws_H = H; u->ws_W = W;
    }
    return SISIC_OK;
}

// time embedding + all time_emb_proj rows for `rows` timesteps (t_vals already on the device)
int time_embed(sisic_unet* u, int rows, hipStream_t s) {
    SISIC_TRY(launch_temb_mlp(u->ctx, u->t_vals, rows, u->d_freqs, u->cfg.n_freqs, u->w1t, u->rawp(u->temb_b1), u->w2t,
                              u->rawp(u->temb_b2), u->hidden, u->temb_act, s));
    SISIC_TRY(launch_linear_t(u->ctx, u->temb_act, rows, u->hidden, u->tproj_wt, u->tproj_b, u->tproj_R, u->tproj, s));
    return SISIC_OK;
}

int run_forward(sisic_unet* u, const float* sample, const float* tproj, int tproj_stride, float* out, int B, int H,
                int W, hipStream_t s, TrainState* tr = nullptr) {
    Fwd f{u, s, B, tproj, tproj_stride, {}};
    f.tr = tr;
    f.gsc = u->gn_scale; f.gsh = u->gn_shift;
    const int rc = f.run(sample, out, H, W);
    if (tr && rc == SISIC_OK) {        // training mode: the tape owns the activations until the backward pass has run
        for (auto& b : f.bufs) tr->bufs.push_back(std::move(b));
        return rc;
    }
    for (auto& b : f.bufs) {           // inference, and error paths: hand everything back
        if (b->p) pool_put(u, b->p);
        if (b->stats) pool_put(u, b->stats);
    }
    if (tr) tr->tape.clear();
    return rc;
}

}  // namespace

namespace sisic {
int unet_pool_get(sisic_unet* u, size_t floats, float** out) { return pool_get(u, floats, out); }
void unet_release_tape(sisic_unet* u) { tape_drop(u, true); }
void unet_pool_put(sisic_unet* u, float* p) { pool_put(u, p); }
int unet_grow(float** p, size_t* have, size_t want) { return grow(p, have, want); }
int unet_check_shape(sisic_unet* u, int B, int H, int W) { return check_shape(u, B, H, W); }
int unet_ensure_rows(sisic_unet* u, size_t t_rows, size_t gn_rows) { return ensure_rows(u, t_rows, gn_rows); }
int unet_stage_upload(sisic_unet* u, const float* src, size_t n, float* dst, hipStream_t s) { return stage_upload(u, src, n, dst, s); }
int unet_prepare_all(sisic_unet* u, hipStream_t s) { return prepare_all(u, s); }
int unet_run_forward(sisic_unet* u, const float* sample, const float* tproj, int tproj_stride, float* out, int B, int H,
                     int W, hipStream_t s, TrainState* tape) {
    return run_forward(u, sample, tproj, tproj_stride, out, B, H, W, s, tape);
}
}  // namespace sisic

extern "C" {

int sisic_unet_create(sisic_ctx* ctx, const sisic_unet_config* cfg, sisic_unet** out) {
    SISIC_REQUIRE(ctx && cfg && out, "unet_create: null argument");
    *out = nullptr;
    SISIC_REQUIRE(cfg->n_blocks >= 1 && cfg->n_blocks <= 8, "unet_create: n_blocks %d", cfg->n_blocks);
    SISIC_REQUIRE(cfg->layers_per_block >= 1, "unet_create: layers_per_block");
    SISIC_REQUIRE(cfg->in_channels > 0 && cfg->out_channels > 0, "unet_create: channels");
    SISIC_REQUIRE(cfg->norm_groups > 0 && cfg->head_dim == 8, "unet_create: head_dim must be 8, groups > 0");
    SISIC_REQUIRE(cfg->freqs && cfg->n_freqs * 2 == cfg->block_out_channels[0], "unet_create: freqs table must hold block_out_channels[0]/2 entries");
    for (int i = 0; i < cfg->n_blocks; ++i)
        SISIC_REQUIRE(cfg->block_out_channels[i] > 0 && cfg->block_out_channels[i] % cfg->norm_groups == 0 &&
                          cfg->block_out_channels[i] % cfg->head_dim == 0,
                      "unet_create: block_out_channels[%d]=%d", i, cfg->block_out_channels[i]);
    auto* u = new sisic_unet();
    u->ctx = ctx;
    u->cfg = *cfg;
    u->freqs.assign(cfg->freqs, cfg->freqs + cfg->n_freqs);
    if (const char* e = std::getenv("SISIC_WINOGRAD")) u->use_winograd = std::atoi(e) != 0;
    if (const char* e = std::getenv("SISIC_FUSED_GN")) u->fuse_gn = std::atoi(e) != 0;
    if (const char* e = std::getenv("SISIC_GRAPH")) u->graph_mode = std::atoi(e) != 0 ? 1 : 0;
    u->cfg.freqs = nullptr;
    const int rc = describe(u);
    if (rc != SISIC_OK) { delete u; return rc; }
    *out = u;
    return SISIC_OK;
}

int sisic_unet_destroy(sisic_unet* u) {
    if (!u) return SISIC_OK;
    (void)sisic_unet_train_end(u);
    (void)hipDeviceSynchronize();
    pool_release_all(u);
    for (float* p : {u->x_work, u->loop_tables, u->tproj_cur})
        if (p) (void)hipFree(p);
    if (u->loop_stream) (void)hipStreamDestroy(u->loop_stream);
    for (auto p : u->owned) (void)hipFree(p);
    for (float* p : {u->raw, u->t_vals, u->temb_act, u->tproj, u->gn_scale, u->gn_shift, u->gn_scale2, u->gn_shift2, u->eps_buf})
        if (p) (void)hipFree(p);
    if (u->stage_host) (void)hipHostFree(u->stage_host);
    for (auto e : u->stage_ev)
        if (e) (void)hipEventDestroy(e);
    delete u;
    return SISIC_OK;
}

int sisic_unet_set_latency_mode(sisic_unet* u, int on) {
    SISIC_REQUIRE(u, "set_latency_mode: null handle");
    u->latency_mode = on != 0;
    return SISIC_OK;
}

int sisic_unet_set_graph_mode(sisic_unet* u, int mode) {
    SISIC_REQUIRE(u && mode >= -1 && mode <= 1, "set_graph_mode: mode must be -1 (follow latency mode), 0 or 1");
    u->graph_mode = mode;
    return SISIC_OK;
}

int64_t sisic_unet_graph_builds(const sisic_unet* u) { return u ? u->loop_builds : 0; }

int64_t sisic_unet_workspace_bytes(const sisic_unet* u) {
    int64_t n = 0;
    if (u) for (const auto& b : u->pool) n += (int64_t)b.bytes;
    return n;
}

int sisic_unet_num_tensors(const sisic_unet* u) { return u ? (int)u->names.size() : 0; }

const char* sisic_unet_tensor_name(const sisic_unet* u, int i) {
    if (!u || i < 0 || i >= (int)u->names.size()) return nullptr;
    return u->names[i].c_str();
}

int sisic_unet_load(sisic_unet* u, int n, const char* const* names, const float* const* host_ptrs,
                    const int64_t* numels) {
    SISIC_REQUIRE(u && names && host_ptrs && numels, "unet_load: null argument");
    SISIC_REQUIRE(n == (int)u->names.size(), "unet_load: state dict has %d tensors, expected %d", n, (int)u->names.size());
    SISIC_HIP(hipSetDevice(u->ctx->device));
    if (!u->raw) {
        void* p = nullptr;
        SISIC_HIP(hipMalloc(&p, u->raw_floats * sizeof(float)));
        u->raw = static_cast<float*>(p);
    }
    SISIC_HIP(hipMemset(u->raw, 0, u->raw_floats * sizeof(float)));
    std::vector<char> seen(u->names.size(), 0);
    for (int i = 0; i < n; ++i) {
        SISIC_REQUIRE(names[i] && host_ptrs[i], "unet_load: entry %d is null", i);
        auto it = u->index.find(names[i]);
        SISIC_REQUIRE(it != u->index.end(), "unet_load: unexpected key '%s'", names[i]);
        const int idx = it->second;
        SISIC_REQUIRE(!seen[idx], "unet_load: duplicate key '%s'", names[i]);
        SISIC_REQUIRE(numels[i] == u->numels[idx], "unet_load: '%s' has %lld elements, expected %lld", names[i],
                      (long long)numels[i], (long long)u->numels[idx]);
        seen[idx] = 1;
        SISIC_HIP(hipMemcpy(u->raw + u->offsets[idx], host_ptrs[i], (size_t)numels[i] * sizeof(float), hipMemcpyHostToDevice));
    }
    // (re)build derived buffers.  A captured sampling step holds the addresses of the packed filters freed below: drop it
    // (a replay after load_state_dict would read freed memory).
    SISIC_HIP(hipDeviceSynchronize());
    loop_graph_drop(u);
    for (auto p : u->owned) (void)hipFree(p);
    u->owned.clear();
    reset_derived(u);
    u->loaded = false;
    SISIC_TRY(prepare_all(u, nullptr));
    SISIC_HIP(hipDeviceSynchronize());
    u->loaded = true;
    return SISIC_OK;
}

int sisic_unet_forward(sisic_unet* u, const float* sample, const int64_t* timesteps, float* out, int B, int H, int W,
                       void* stream) {
    SISIC_REQUIRE(u && sample && timesteps && out, "unet_forward: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    SISIC_TRY(check_shape(u, B, H, W));
    SISIC_TRY(ensure_rows(u, (size_t)B, (size_t)B));
    std::vector<float> tv(B);
    bool uniform = true;
    for (int b = 0; b < B; ++b) {
        tv[b] = (float)timesteps[b];
        uniform = uniform && timesteps[b] == timesteps[0];
    }
    const int rows = uniform ? 1 : B;
    SISIC_TRY(stage_upload(u, tv.data(), (size_t)rows, u->t_vals, s));
    SISIC_TRY(time_embed(u, rows, s));
    return run_forward(u, sample, u->tproj, uniform ? 0 : u->tproj_R, out, B, H, W, s);
}

// One denoising step with every per-step parameter selected on the device (elementwise.hip, LoopState): identical
// launches for every step, so that a captured step can be replayed.
static int loop_step(sisic_unet* u, int B, int H, int W, size_t n, float clip, hipStream_t s) {
    void* state = u->loop_tables;
    const float* coef_dev = u->loop_tables + 4;
    const int* zrow_dev = reinterpret_cast<const int*>(u->loop_tables + 4 + 5 * 1000);
    SISIC_TRY(launch_loop_select_row(u->ctx, u->tproj, u->tproj_R, state, u->tproj_cur, s));
    SISIC_TRY(run_forward(u, u->x_work, u->tproj_cur, 0, u->eps_buf, B, H, W, s));
    SISIC_TRY(launch_ddpm_step_indexed(u->ctx, u->eps_buf, u->x_work, (int64_t)n, state, coef_dev, zrow_dev, clip, s));
    return launch_loop_advance(u->ctx, state, s);
}

// The loop as ONE captured step replayed T-1 times (hipGraph): at batch 1 a step is ~190 launches of 5-20 us kernels and
// the host cannot issue them as fast as the GPU retires them (measured: 3.0 ms of kernels in a 4.5 ms step).
static int sample_graph(sisic_unet* u, float* x, int B, int H, int W, int T, const float* coef, float clip, const float* noise,
                        float* traj, const int* traj_row, const volatile int* cancel, int* steps_done, hipStream_t caller) {
    const int C = u->cfg.in_channels;
    const size_t n = (size_t)B * C * H * W;
    SISIC_REQUIRE(T <= 1000, "sample: at most 1000 steps per call");
    // the replayed launches write into the pool blocks they were captured with; a tape recorded since then may own some of
    // them: the tape does not survive a graph-replayed run (sisic_unet_backward then answers SISIC_ESTATE)
    tape_drop(u, true);
    hipStream_t s = caller;
    if (!s) {                              // the legacy default stream cannot be captured: a blocking stream of our own,
        if (!u->loop_stream) SISIC_HIP(hipStreamCreate(&u->loop_stream));     // implicitly ordered with the default stream
        s = u->loop_stream;
    }
    SISIC_TRY(grow(&u->x_work, &u->x_work_cap, n));
    SISIC_TRY(grow(&u->tproj_cur, &u->tproj_cur_cap, (size_t)u->tproj_R));
    SISIC_TRY(grow(&u->loop_tables, &u->loop_tables_cap, (size_t)4 + 5 * 1000 + 1000));
    // tables of this call: {step = 0, noise base}, coefficients, noise row per step (-1: the step adds no noise)
    std::vector<float> tab(4 + 5 * 1000 + 1000, 0.0f);
    {
        int step0 = 0;
        std::memcpy(&tab[0], &step0, sizeof(int));
        std::memcpy(&tab[2], &noise, sizeof(noise));
        std::memcpy(&tab[4], coef, (size_t)T * 5 * sizeof(float));
        int* zr = reinterpret_cast<int*>(&tab[4 + 5 * 1000]);
        int zi = 0;
        for (int i = 0; i < T; ++i) zr[i] = (noise && coef[(size_t)i * 5 + 4] != 0.0f) ? zi++ : -1;
    }
    SISIC_TRY(stage_upload(u, tab.data(), tab.size(), u->loop_tables, s));
    SISIC_HIP(hipMemcpyAsync(u->x_work, x, n * sizeof(float), hipMemcpyDeviceToDevice, s));

    auto after_step = [&](int i) -> int {
        const int row = traj_row ? traj_row[i] : i;          // (kept frames only: XAI.py:751-757 save_indices)
        if (traj && row >= 0) SISIC_HIP(hipMemcpyAsync(traj + (size_t)row * n, u->x_work, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (steps_done) *steps_done = i + 1;
        return SISIC_OK;
    };
    int rc = SISIC_OK;
    int i = 0;
    auto cancelled = [&](int at) -> bool {
        if (!cancel) return false;
        if ((at & 7) == 0) (void)hipStreamSynchronize(s);
        return *cancel != 0;
    };
    if (cancelled(0)) rc = SISIC_ECANCEL;
    // every address baked into the captured launches: the pool and the scratch buffers are sized by the first eager step at a
    // shape, the tables by ensure_rows / grow above
    auto reusable = [&]() -> bool {
        const uint64_t gen = u->ctx->scratch_generation.load();
        const void* ptrs[5] = {u->tproj, u->eps_buf, u->x_work, u->loop_tables, u->tproj_cur};
        bool ok = u->loop_valid && u->loop_key.B == B && u->loop_key.H == H && u->loop_key.W == W &&
                  u->loop_key.clip == clip && u->loop_key.s == s && u->loop_key.latency == u->latency_mode &&
                  u->loop_key.gen == gen;
        for (int k = 0; k < 5; ++k) ok = ok && u->loop_key.ptrs[k] == ptrs[k];
        return ok;
    };
    if (rc == SISIC_OK && !reusable()) {
        // step 0 eagerly: sizes the pool and every scratch buffer, opts the kernels in to their LDS sizes.  (A call that finds
        // its graph -- every call after the first at a shape -- replays from step 0: the eager step is ~190 launches, twice
        // the time of a replayed one at batch 1.)
        rc = loop_step(u, B, H, W, n, clip, s);
        if (rc == SISIC_OK) rc = after_step(0);
        i = 1;
    }
    if (rc == SISIC_OK && i < T) {
        if (!reusable()) {
            const uint64_t gen = u->ctx->scratch_generation.load();
            const void* ptrs[5] = {u->tproj, u->eps_buf, u->x_work, u->loop_tables, u->tproj_cur};
            loop_graph_drop(u);
            SISIC_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
            const int crc = loop_step(u, B, H, W, n, clip, s);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(s, &g);
            if (crc != SISIC_OK || e != hipSuccess || !g) {
                if (g) (void)hipGraphDestroy(g);
                if (crc == SISIC_OK) set_error("sample: stream capture failed: %s", hipGetErrorString(e));
                return crc != SISIC_OK ? crc : SISIC_EHIP;
            }
            u->loop_graph = g;
            SISIC_HIP(hipGraphInstantiate(&u->loop_exec, g, nullptr, nullptr, 0));
            u->loop_key.B = B; u->loop_key.H = H; u->loop_key.W = W; u->loop_key.clip = clip; u->loop_key.s = s;
            u->loop_key.latency = u->latency_mode; u->loop_key.gen = gen;
            for (int k = 0; k < 5; ++k) u->loop_key.ptrs[k] = ptrs[k];
            u->loop_valid = true;
            u->loop_builds += 1;
        }
        for (; i < T; ++i) {
            if (cancelled(i)) { rc = SISIC_ECANCEL; break; }
            SISIC_HIP(hipGraphLaunch(u->loop_exec, s));
            SISIC_TRY(after_step(i));
        }
    }
    SISIC_HIP(hipMemcpyAsync(x, u->x_work, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (s != caller) SISIC_HIP(hipStreamSynchronize(s));      // hand the result back to the default stream's order
    if (rc == SISIC_ECANCEL) {
        SISIC_HIP(hipStreamSynchronize(s));
        set_error("sample: cancelled after %d of %d steps", steps_done ? *steps_done : i, T);
    }
    return rc;
}

int sisic_sample(sisic_unet* u, float* x, int B, int H, int W, int T, const int64_t* timesteps, const float* coef,
                 float clip, const float* noise, float* traj, uint8_t* out_u8, const volatile int* cancel,
                 int* steps_done, void* stream) {
    return sisic_sample_frames(u, x, B, H, W, T, timesteps, coef, clip, noise, traj, nullptr, out_u8, cancel, steps_done, stream);
}

int sisic_sample_frames(sisic_unet* u, float* x, int B, int H, int W, int T, const int64_t* timesteps, const float* coef,
                        float clip, const float* noise, float* traj, const int* traj_row, uint8_t* out_u8,
                        const volatile int* cancel, int* steps_done, void* stream) {
    SISIC_REQUIRE(u && x && timesteps && coef && T > 0, "sample: null argument");
    SISIC_REQUIRE(!traj_row || traj, "sample: traj_row without a trajectory buffer");
    if (traj_row)
        for (int i = 0; i < T; ++i) SISIC_REQUIRE(traj_row[i] >= -1, "sample: traj_row[%d] = %d (a row of traj, or -1)", i, traj_row[i]);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (steps_done) *steps_done = 0;
    SISIC_TRY(check_shape(u, B, H, W));
    // per-step tables for 1000 rows from the first call on (17 MB): a longer run after a shorter one then never moves the
    // time-embedding table, so the captured step (which holds its address) survives a change of T
    SISIC_TRY(ensure_rows(u, (size_t)std::max(T, 1000), (size_t)B));
    const int C = u->cfg.in_channels;
    SISIC_REQUIRE(u->cfg.out_channels == C, "sample: in/out channels differ");
    const size_t n = (size_t)B * C * H * W;
    SISIC_TRY(grow(&u->eps_buf, &u->eps_floats, n));

    // every step's time embedding and time_emb_proj rows in one batch before the loop
    std::vector<float> tv(T);
    for (int i = 0; i < T; ++i) tv[i] = (float)timesteps[i];
    SISIC_TRY(stage_upload(u, tv.data(), (size_t)T, u->t_vals, s));
    SISIC_TRY(time_embed(u, T, s));

    const bool use_graph = (u->graph_mode < 0 ? u->latency_mode : u->graph_mode != 0) && !u->ctx->profiling && T >= 4 && T <= 1000;
    if (use_graph) {
        if (!s) SISIC_HIP(hipStreamSynchronize(s));           // the embeddings above ran on the default stream
        const int rc = sample_graph(u, x, B, H, W, T, coef, clip, noise, traj, traj_row, cancel, steps_done, s);
        if (rc != SISIC_OK) return rc;
        if (out_u8) SISIC_TRY(launch_denorm_u8(u->ctx, x, out_u8, B, C, H, W, s));
        return SISIC_OK;
    }

    size_t zi = 0;
    for (int i = 0; i < T; ++i) {
        if (cancel) {
            if ((i & 7) == 0) SISIC_HIP(hipStreamSynchronize(s));   // bound the run-ahead so a stop request takes effect
            if (*cancel) {
                SISIC_HIP(hipStreamSynchronize(s));
                set_error("sample: cancelled after %d of %d steps", i, T);
                return SISIC_ECANCEL;
            }
        }
        SISIC_TRY(run_forward(u, x, u->tproj + (size_t)i * u->tproj_R, 0, u->eps_buf, B, H, W, s));
        const float* c = coef + (size_t)i * 5;
        const float* z = nullptr;
        if (noise && c[4] != 0.0f) z = noise + (zi++) * n;
        SISIC_TRY(launch_ddpm_step(u->ctx, u->eps_buf, x, z, x, (int64_t)n, c[0], c[1], c[2], c[3], c[4], clip, s));
        const int row = traj_row ? traj_row[i] : i;
        if (traj && row >= 0) SISIC_HIP(hipMemcpyAsync(traj + (size_t)row * n, x, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (steps_done) *steps_done = i + 1;
    }
    if (out_u8) SISIC_TRY(launch_denorm_u8(u->ctx, x, out_u8, B, C, H, W, s));
    return SISIC_OK;
}

}  // extern "C"
