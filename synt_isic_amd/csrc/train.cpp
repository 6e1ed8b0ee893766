// train.cpp -- one UNet training step on the HIP kernels.
//
// Mirrors the loop body of diffusion/train_diffusion.py:211-240
//     noisy = scheduler.add_noise(images, noise, timesteps)            sisic_add_noise
//     noise_pred = model(noisy, timesteps).sample                      sisic_unet_train_forward  (records a tape)
//     loss = F.mse_loss(noise_pred, noise)                             sisic_mse_loss            (loss and d loss / d pred)
//     scaler.scale(loss).backward()                                    sisic_unet_backward       (walks the tape backwards)
//     scaler.step(optimizer); scaler.update()                          sisic_unet_optimizer_step (unscale, inf check, Adam)
// and sisic_unet_train_step runs the five in one call on one stream.  The forward is the inference executor of unet.cpp
// in tape mode: same kernels (Winograd / direct MFMA convolutions with fused GroupNorm+SiLU prologues, GroupNorm
// statistics from convolution epilogues, attention), nothing released, every GroupNorm's scale/shift/(mean, rstd) kept.
// Backward per convolution: bias / time-embedding sums, backward-weight (conv_wgrad_kernel), backward-data as a
// forward convolution with transposed filters, then GroupNorm+SiLU backward into the input's gradient.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "train.h"
#include "unet_internal.h"

using namespace sisic;

namespace {

float* grad_of(sisic_unet* u, int idx) { return u->train->grad + u->offsets[idx]; }

// filters of the backward-data convolutions, rebuilt from the raw arena (after load and after every optimizer step)
int prepare_backward_conv(sisic_unet* u, ConvW& c, hipStream_t s) {
    if (c.k == 0) return SISIC_OK;
    const int kk = c.k * c.k;
    auto alloc = [&](size_t floats, float** p) -> int {
        if (*p) return SISIC_OK;
        void* q = nullptr;
        SISIC_HIP(hipMalloc(&q, std::max<size_t>(floats, 4) * sizeof(float)));
        u->owned.push_back(static_cast<float*>(q));
        *p = static_cast<float*>(q);
        return SISIC_OK;
    };
    SISIC_TRY(alloc((size_t)c.cout * c.cin * kk, &c.raw_t));
    SISIC_TRY(launch_transpose_flip(u->ctx, u->rawp(c.w_idx), c.cout, c.cin, kk, c.raw_t, s));
    SISIC_TRY(alloc((size_t)sisic_conv_packed_numel(c.cin, c.cout, c.k), &c.packed_t));
    SISIC_TRY(launch_conv_pack(u->ctx, c.raw_t, c.cin, c.cout, c.k, c.packed_t, s));
    if (c.k == 3 && !c.strided && c.cin > 4) {
        SISIC_TRY(alloc((size_t)winograd_packed_numel(c.cin, c.cout), &c.wino_t));
        SISIC_TRY(launch_winograd_pack(u->ctx, c.raw_t, c.cin, c.cout, c.wino_t, s));
    }
    return SISIC_OK;
}

// The job tables of repack.hip: everything prepare_all (unet.cpp) and prepare_backward_weights (below) derive from the raw
// arena, in dependency order.  Built after those two have run once (they allocate); the destinations do not move afterwards.
int build_repack_plan(sisic_unet* u) {
    TrainState* tr = u->train.get();
    std::vector<PackJob> ph[3];
    const int nin = 2 * u->cfg.n_freqs;
    ph[0].push_back(pack_job_transpose2d(u->rawp(u->temb_w1), u->hidden, nin, u->w1t, u->hidden, 0));
    ph[0].push_back(pack_job_transpose2d(u->rawp(u->temb_w2), u->hidden, u->hidden, u->w2t, u->hidden, 0));
    for (ConvW* c : unet_convs(u)) {
        if (c->k == 0) continue;
        const float* w = u->rawp(c->w_idx);
        SISIC_REQUIRE(c->packed, "repack plan: forward filters are not prepared");
        ph[0].push_back(pack_job_conv(w, c->cout, c->cin, c->k, c->packed));
        if (c->wino) {
            ph[0].push_back(pack_job_wino_first(w, c->cout, c->cin, c->wino));
            ph[1].push_back(pack_job_wino_wide(c->cout, c->cin, c->wino));
            ph[1].push_back(pack_job_wino_bf3(c->cout, c->cin, c->wino));
        }
        if (c == &u->conv_in) continue;                       // the network input needs no gradient
        SISIC_REQUIRE(c->raw_t && c->packed_t, "repack plan: backward filters are not prepared");
        ph[0].push_back(pack_job_flip(w, c->cout, c->cin, c->k * c->k, c->raw_t));
        ph[1].push_back(pack_job_conv(c->raw_t, c->cin, c->cout, c->k, c->packed_t));
        if (c->wino_t) {
            ph[1].push_back(pack_job_wino_first(c->raw_t, c->cin, c->cout, c->wino_t));
            ph[2].push_back(pack_job_wino_wide(c->cin, c->cout, c->wino_t));
            ph[2].push_back(pack_job_wino_bf3(c->cin, c->cout, c->wino_t));
        }
    }
    for (ResnetW* r : unet_resnets(u)) {
        ph[0].push_back(pack_job_transpose2d(u->rawp(r->temb_w_idx), r->cout, u->hidden, u->tproj_wt, u->tproj_R, r->temb_off));
        ph[0].push_back(pack_job_copy(u->rawp(r->temb_b_idx), u->tproj_b + r->temb_off, (size_t)r->cout));
    }
    for (AttnW* a : unet_attns(u)) {
        const int c = a->c;
        SISIC_REQUIRE(a->qkv_cat && a->qkv_packed && a->qkv_bias && a->qkv_raw_t && a->qkv_packed_t, "repack plan: attention filters are not prepared");
        const int wi[3] = {a->q_w, a->k_w, a->v_w}, bi[3] = {a->q_b, a->k_b, a->v_b};
        for (int i = 0; i < 3; ++i) {
            ph[0].push_back(pack_job_copy(u->rawp(wi[i]), a->qkv_cat + (size_t)i * c * c, (size_t)c * c));
            ph[0].push_back(pack_job_copy(u->rawp(bi[i]), a->qkv_bias + (size_t)i * c, (size_t)c));
        }
        ph[1].push_back(pack_job_conv(a->qkv_cat, 3 * c, c, 1, a->qkv_packed));
        ph[1].push_back(pack_job_flip(a->qkv_cat, 3 * c, c, 1, a->qkv_raw_t));
        ph[2].push_back(pack_job_conv(a->qkv_raw_t, c, 3 * c, 1, a->qkv_packed_t));
    }
    for (int p = 0; p < 3; ++p) {
        int blocks = 0;
        for (PackJob& j : ph[p]) { j.first_block = blocks; blocks += pack_job_blocks(j); }
        if (tr->repack_dev[p]) { (void)hipFree(tr->repack_dev[p]); tr->repack_dev[p] = nullptr; }
        SISIC_HIP(hipMalloc(&tr->repack_dev[p], std::max<size_t>(ph[p].size(), 1) * sizeof(PackJob)));
        SISIC_HIP(hipMemcpy(tr->repack_dev[p], ph[p].data(), ph[p].size() * sizeof(PackJob), hipMemcpyHostToDevice));
        tr->repack_jobs[p] = (int)ph[p].size();
        tr->repack_blocks[p] = blocks;
    }
    tr->repack_ready = true;
    return SISIC_OK;
}

int run_repack_plan(sisic_unet* u, hipStream_t s) {
    TrainState* tr = u->train.get();
    for (int p = 0; p < 3; ++p)
        SISIC_TRY(launch_pack_batch(u->ctx, static_cast<const PackJob*>(tr->repack_dev[p]), tr->repack_jobs[p], tr->repack_blocks[p], s));
    return SISIC_OK;
}

int prepare_backward_weights(sisic_unet* u, hipStream_t s) {
    for (ConvW* c : unet_convs(u))
        if (c != &u->conv_in) SISIC_TRY(prepare_backward_conv(u, *c, s));     // the network input needs no gradient
    for (AttnW* a : unet_attns(u)) {
        const int c = a->c;
        auto alloc = [&](size_t floats, float** p) -> int {
            if (*p) return SISIC_OK;
            void* q = nullptr;
            SISIC_HIP(hipMalloc(&q, floats * sizeof(float)));
            u->owned.push_back(static_cast<float*>(q));
            *p = static_cast<float*>(q);
            return SISIC_OK;
        };
        SISIC_TRY(alloc((size_t)3 * c * c, &a->qkv_raw_t));
        SISIC_TRY(launch_transpose_flip(u->ctx, a->qkv_cat, 3 * c, c, 1, a->qkv_raw_t, s));
        SISIC_TRY(alloc((size_t)sisic_conv_packed_numel(c, 3 * c, 1), &a->qkv_packed_t));
        SISIC_TRY(launch_conv_pack(u->ctx, a->qkv_raw_t, c, 3 * c, 1, a->qkv_packed_t, s));
    }
    return SISIC_OK;
}

void release_tape(sisic_unet* u) { unet_release_tape(u); }

struct Bwd {
    sisic_unet* u;
    TrainState* tr;
    hipStream_t s;
    int B;

    static size_t arena_floats(size_t n) { return (n + 63) & ~size_t(63); }       // 256-byte aligned carve-outs

    // gradient buffer of an activation: carved from the arena that run() zero-filled in one go (one fill instead of one
    // per activation: 85 launches of ~6 us at batch 32)
    int grad(Buf* b, float** out) {
        auto it = tr->buf_grad.find(b);
        if (it != tr->buf_grad.end()) { *out = it->second; return SISIC_OK; }
        const size_t n = arena_floats((size_t)B * b->C * b->H * b->W);
        SISIC_REQUIRE(tr->garena_used + n <= tr->garena_cap, "backward: gradient arena exhausted");
        float* g = tr->garena + tr->garena_used;
        tr->garena_used += n;
        tr->buf_grad[b] = g;
        *out = g;
        return SISIC_OK;
    }

    int conv_op(const TapeOp& op, const float* dout) {
        const ConvW& c = op.w;
        const int Cin = op.c0 + op.c1;
        const int Hc = op.H << (op.ups ? 1 : 0), Wc = op.W << (op.ups ? 1 : 0);
        const int pad = c.k / 2;
        const int Ho = (Hc + 2 * pad - c.k) / op.stride + 1, Wo = (Wc + 2 * pad - c.k) / op.stride + 1;
        const float* dy = dout;
        if (op.out) {
            float* g = nullptr;
            SISIC_TRY(grad(op.out, &g));
            dy = g;
        }
        SISIC_REQUIRE(dy, "backward: no gradient for the network output");
        // ---- residual input: the gradient passes through unchanged
        if (op.residual) {
            float* gr = nullptr;
            SISIC_TRY(grad(op.residual, &gr));
            SISIC_TRY(launch_add_inplace(u->ctx, gr, dy, (size_t)B * c.cout * Ho * Wo, s));
        }
        // ---- bias and time-embedding projection: sums of dy over the pixels of each (sample, channel) plane, one launch
        if (op.qkv_of) {
            SISIC_TRY(launch_bias_grad(u->ctx, dy, B, c.cout, Ho * Wo, grad_of(u, op.qkv_of->q_b), grad_of(u, op.qkv_of->k_b),
                                       grad_of(u, op.qkv_of->v_b), op.qkv_of->c, nullptr, 0, s));
        } else {
            SISIC_TRY(launch_bias_grad(u->ctx, dy, B, c.cout, Ho * Wo, grad_of(u, c.b_idx), nullptr, nullptr, 0,
                                       op.temb_off >= 0 ? tr->dtproj + op.temb_off : nullptr, u->tproj_R, s));
        }
        // ---- weights
        {
            WgradArgs a;
            a.in0 = op.in0_ptr; a.in1 = op.in1_ptr; a.c0 = op.c0; a.c1 = op.c1; a.B = B; a.Hin = op.H; a.Win = op.W;
            a.ups = op.ups; a.ksize = c.k; a.stride = op.stride;
            a.gn_scale = op.gn_scale; a.gn_shift = op.gn_shift; a.gn_silu = op.silu ? 1 : 0;
            a.dy = dy; a.Cout = c.cout;
            const size_t need = conv_wgrad_scratch_floats(a);
            SISIC_TRY(unet_grow(&tr->wgrad_part, &tr->wgrad_part_cap, need));
            if (op.qkv_of) {            // [3C, C] leaves the reduction as the three projections' gradients
                a.dw = grad_of(u, op.qkv_of->q_w); a.dw1 = grad_of(u, op.qkv_of->k_w); a.dw2 = grad_of(u, op.qkv_of->v_w);
            } else {
                a.dw = grad_of(u, c.w_idx);
            }
            SISIC_TRY(launch_conv_wgrad(u->ctx, a, tr->wgrad_part, need, s));
        }
        // ---- data
        if (!op.in0) return SISIC_OK;                          // the network input
        const float* packed_t = op.qkv_of ? op.qkv_of->qkv_packed_t : c.packed_t;
        SISIC_REQUIRE(packed_t, "backward: the backward-data filters have not been prepared");
        const size_t da_n = (size_t)B * Cin * Hc * Wc;
        SISIC_TRY(unet_grow(&tr->scratch, &tr->scratch_cap, da_n));
        float* da = tr->scratch;
        float* g0 = nullptr;
        float* g1 = nullptr;
        SISIC_TRY(grad(op.in0, &g0));
        if (op.in1) SISIC_TRY(grad(op.in1, &g1));
        // an input that is neither normalised, nor upsampled, nor a concatenation receives its gradient straight from the
        // convolution's epilogue: out = conv(dy) + residual with residual = out = the input's gradient buffer (every element is
        // read and then written by the same thread) -- no scratch tensor, no accumulate launch; g + y in place of g += y: same bits
        const bool in_place = !op.norm && !op.ups && !op.in1;
        {
            sisic_conv_args a{};
            a.in0 = dy; a.c0 = c.cout; a.B = B; a.Hin = Ho; a.Win = Wo;
            a.ksize = c.k; a.stride = 1; a.upsample = op.stride == 2 ? 2 : 0;
            a.w_packed = packed_t; a.Cout = Cin;
            a.w_winograd = (op.stride == 1 && !op.qkv_of && u->use_winograd) ? c.wino_t : nullptr;
            a.out = da;
            if (in_place) { a.out = g0; a.residual = g0; }
            if (u->latency_mode) {           // the forward's small-batch tile choices (unet.cpp, Fwd::conv) for the data gradient
                if (c.k == 3 && op.stride == 1 && a.w_winograd && Cin > 4 && Ho >= 12 && Wo >= 12) a.tile_cfg = 79;
                else if (c.k == 1) a.tile_cfg = 22;
            }
            SISIC_TRY(launch_conv2d(u->ctx, a, s));
        }
        if (in_place) return SISIC_OK;
        if (op.norm) {
            float* sums = tr->small + (size_t)B * c.cout;      // [2][B][Cin]
            SISIC_TRY(launch_gn_bwd(u->ctx, da, op.in0_ptr, op.c0, op.in1_ptr, op.c1, B, op.H * op.W, u->cfg.norm_groups,
                                    op.gn_scale, op.gn_shift, op.gn_mr, op.norm->gamma, op.silu ? 1 : 0, sums, g0, g1,
                                    grad_of(u, op.norm->w_idx), grad_of(u, op.norm->b_idx), s));
        } else if (op.ups) {
            SISIC_TRY(launch_accum_pool2(u->ctx, da, B * Cin, op.H, op.W, g0, s));
        } else {
            SISIC_TRY(launch_accum_split(u->ctx, da, B, Cin, op.H * op.W, g0, op.c0, g1, op.c1, s));
        }
        return SISIC_OK;
    }

    int attn_op(const TapeOp& op) {
        float* dO = nullptr;
        float* dqkv = nullptr;
        SISIC_TRY(grad(op.o, &dO));
        SISIC_TRY(grad(op.qkv, &dqkv));                        // zero-filled; the kernel overwrites every element
        return launch_attention_bwd(u->ctx, op.qkv->p, op.o->p, dO, dqkv, B, op.C, op.N, u->cfg.head_dim, s);
    }

    // time embedding: tproj = time_emb_proj(ta), ta = silu(t2), t2 = linear_2(silu(h1)), h1 = linear_1(emb)
    int time_embedding() {
        const int R = u->tproj_R, Hd = u->hidden, nin = 2 * u->cfg.n_freqs;
        const size_t need = (size_t)R * Hd + R + (size_t)6 * B * Hd;
        SISIC_TRY(unet_grow(&tr->wgrad_part, &tr->wgrad_part_cap, need));
        float* dWf = tr->wgrad_part;                // [R][Hd]
        float* dbf = dWf + (size_t)R * Hd;          // [R]
        float* dta = dbf + R;                       // [B][Hd]
        float* dt2 = dta + (size_t)B * Hd;
        float* a1 = dt2 + (size_t)B * Hd;
        float* da1 = a1 + (size_t)B * Hd;
        float* dh1 = da1 + (size_t)B * Hd;
        float* ones = dh1 + (size_t)B * Hd;         // scratch [B][Hd]
        // fused projection of the 22 residual blocks
        SISIC_TRY(launch_linear_wgrad(u->ctx, tr->dtproj, R, u->temb_act, B, R, Hd, dWf, s));
        SISIC_TRY(launch_col_sums(u->ctx, tr->dtproj, B, R, R, dbf, 0, s));
        // the 22 blocks' slices of the fused gradient to their own tensors: ONE launch over a job table (44 device copies before)
        if (!tr->scatter_dev || tr->scatter_src != dWf) {
            std::vector<PackJob> jobs;
            for (ResnetW* r : unet_resnets(u)) {
                jobs.push_back(pack_job_copy(dWf + (size_t)r->temb_off * Hd, grad_of(u, r->temb_w_idx), (size_t)r->cout * Hd));
                jobs.push_back(pack_job_copy(dbf + r->temb_off, grad_of(u, r->temb_b_idx), (size_t)r->cout));
            }
            int blocks = 0;
            for (PackJob& j : jobs) { j.first_block = blocks; blocks += pack_job_blocks(j); }
            SISIC_HIP(hipStreamSynchronize(s));          // (a table in use is not replaced under a running launch)
            if (tr->scatter_dev) { (void)hipFree(tr->scatter_dev); tr->scatter_dev = nullptr; }
            SISIC_HIP(hipMalloc(&tr->scatter_dev, std::max<size_t>(jobs.size(), 1) * sizeof(PackJob)));
            SISIC_HIP(hipMemcpy(tr->scatter_dev, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
            tr->scatter_jobs = (int)jobs.size(); tr->scatter_blocks = blocks; tr->scatter_src = dWf;
        }
        SISIC_TRY(launch_pack_batch(u->ctx, static_cast<const PackJob*>(tr->scatter_dev), tr->scatter_jobs, tr->scatter_blocks, s));
        // d ta[b][k] = sum_r dtproj[b][r] * Wfused[r][k]; the fused weight is stored transposed: tproj_wt[k][r]
        SISIC_TRY(launch_linear_dgrad(u->ctx, tr->dtproj, R, u->tproj_wt, B, R, Hd, dta, s, /*w_is_transposed=*/1));
        SISIC_TRY(launch_silu_bwd(u->ctx, dta, tr->t2, (size_t)B * Hd, dt2, s));
        // linear_2: t2 = a1 W2^T + b2, a1 = silu(h1)
        SISIC_TRY(launch_silu_fwd(u->ctx, tr->h1, (size_t)B * Hd, a1, s));
        SISIC_TRY(launch_linear_wgrad(u->ctx, dt2, Hd, a1, B, Hd, Hd, grad_of(u, u->temb_w2), s));
        SISIC_TRY(launch_col_sums(u->ctx, dt2, B, Hd, Hd, grad_of(u, u->temb_b2), 0, s));
        SISIC_TRY(launch_linear_dgrad(u->ctx, dt2, Hd, u->rawp(u->temb_w2), B, Hd, Hd, da1, s, 0));
        SISIC_TRY(launch_silu_bwd(u->ctx, da1, tr->h1, (size_t)B * Hd, dh1, s));
        // linear_1: h1 = emb W1^T + b1
        SISIC_TRY(launch_linear_wgrad(u->ctx, dh1, Hd, tr->emb, B, Hd, nin, grad_of(u, u->temb_w1), s));
        SISIC_TRY(launch_col_sums(u->ctx, dh1, B, Hd, Hd, grad_of(u, u->temb_b1), 0, s));
        (void)ones;
        return SISIC_OK;
    }

    int run(const float* dout) {
        SISIC_HIP(hipMemsetAsync(tr->dtproj, 0, (size_t)B * u->tproj_R * sizeof(float), s));
        size_t total = 0;                        // every activation of the tape may receive a gradient
        for (auto& b : tr->bufs) total += arena_floats((size_t)B * b->C * b->H * b->W);
        SISIC_TRY(unet_grow(&tr->garena, &tr->garena_cap, total));
        tr->garena_used = 0;
        SISIC_HIP(hipMemsetAsync(tr->garena, 0, total * sizeof(float), s));
        for (auto it = tr->tape.rbegin(); it != tr->tape.rend(); ++it) {
            if (it->kind == TapeOp::CONV) SISIC_TRY(conv_op(*it, dout));
            else SISIC_TRY(attn_op(*it));
        }
        return time_embedding();
    }
};

int require_train(sisic_unet* u, const char* what) {
    SISIC_REQUIRE(u, "%s: null handle", what);
    if (!u->loaded || !u->train) {
        set_error("%s: call sisic_unet_load and sisic_unet_train_begin first", what);
        return SISIC_ESTATE;
    }
    return SISIC_OK;
}

}  // namespace

extern "C" {

int sisic_unet_train_begin(sisic_unet* u) {
    SISIC_REQUIRE(u, "train_begin: null handle");
    if (!u->loaded) {
        set_error("train_begin: load the weights first");
        return SISIC_ESTATE;
    }
    SISIC_HIP(hipSetDevice(u->ctx->device));
    if (!u->train) {
        auto tr = std::make_unique<TrainState>();
        const size_t bytes = u->raw_floats * sizeof(float);
        for (float** p : {&tr->grad, &tr->adam_m, &tr->adam_v}) {
            void* q = nullptr;
            SISIC_HIP(hipMalloc(&q, bytes));
            *p = static_cast<float*>(q);
        }
        void* q = nullptr;
        SISIC_HIP(hipMalloc(&q, 4 * sizeof(float)));
        tr->loss_dev = static_cast<float*>(q);
        SISIC_HIP(hipMalloc(&q, sizeof(int)));
        tr->flag_dev = static_cast<int*>(q);
        SISIC_HIP(hipMalloc(&q, 2048 * sizeof(float)));
        tr->mse_part = static_cast<float*>(q);
        u->train = std::move(tr);
    }
    TrainState* tr = u->train.get();
    const size_t bytes = u->raw_floats * sizeof(float);
    SISIC_HIP(hipMemset(tr->grad, 0, bytes));
    SISIC_HIP(hipMemset(tr->adam_m, 0, bytes));
    SISIC_HIP(hipMemset(tr->adam_v, 0, bytes));
    tr->step = 0;
    release_tape(u);
    SISIC_TRY(prepare_backward_weights(u, nullptr));
    SISIC_HIP(hipDeviceSynchronize());
    return SISIC_OK;
}

int sisic_unet_train_end(sisic_unet* u) {
    if (!u || !u->train) return SISIC_OK;
    (void)hipDeviceSynchronize();
    release_tape(u);
    TrainState* tr = u->train.get();
    for (float* p : {tr->grad, tr->adam_m, tr->adam_v, tr->emb, tr->h1, tr->t2, tr->dtproj, tr->garena, tr->wgrad_part, tr->scratch,
                     tr->small, tr->loss_dev, tr->mse_part})
        if (p) (void)hipFree(p);
    if (tr->scatter_dev) (void)hipFree(tr->scatter_dev);
    for (void* p : tr->repack_dev)
        if (p) (void)hipFree(p);
    if (tr->flag_dev) (void)hipFree(tr->flag_dev);
    u->train.reset();
    return SISIC_OK;
}

int sisic_unet_zero_grad(sisic_unet* u, void* stream) {
    SISIC_TRY(require_train(u, "zero_grad"));
    SISIC_HIP(hipMemsetAsync(u->train->grad, 0, u->raw_floats * sizeof(float), static_cast<hipStream_t>(stream)));
    return SISIC_OK;
}

int sisic_unet_train_forward(sisic_unet* u, const float* sample, const int64_t* timesteps, float* out, int B, int H, int W,
                             void* stream) {
    SISIC_TRY(require_train(u, "train_forward"));
    SISIC_REQUIRE(sample && timesteps && out, "train_forward: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    TrainState* tr = u->train.get();
    release_tape(u);                                       // a forward without a backward: drop the old tape
    SISIC_TRY(unet_check_shape(u, B, H, W));
    SISIC_TRY(unet_ensure_rows(u, (size_t)B, (size_t)B));
    const int Hd = u->hidden, nin = 2 * u->cfg.n_freqs;
    SISIC_TRY(unet_grow(&tr->emb, &tr->emb_cap, (size_t)B * nin));
    SISIC_TRY(unet_grow(&tr->h1, &tr->h1_cap, (size_t)B * Hd));
    SISIC_TRY(unet_grow(&tr->t2, &tr->t2_cap, (size_t)B * Hd));
    SISIC_TRY(unet_grow(&tr->dtproj, &tr->dtproj_cap, (size_t)B * u->tproj_R));
    SISIC_TRY(unet_grow(&tr->small, &tr->small_cap, train_small_floats(u, B)));
    std::vector<float> tv(B);
    for (int b = 0; b < B; ++b) tv[b] = (float)timesteps[b];
    SISIC_TRY(unet_stage_upload(u, tv.data(), (size_t)B, u->t_vals, s));
    // per-sample rows always (the training timesteps differ per image, train_diffusion.py:216), intermediates kept
    SISIC_TRY(launch_temb_mlp(u->ctx, u->t_vals, B, u->d_freqs, u->cfg.n_freqs, u->w1t, u->rawp(u->temb_b1), u->w2t,
                              u->rawp(u->temb_b2), Hd, u->temb_act, s, tr->emb, tr->h1, tr->t2));
    SISIC_TRY(launch_linear_t(u->ctx, u->temb_act, B, Hd, u->tproj_wt, u->tproj_b, u->tproj_R, u->tproj, s));
    tr->B = B; tr->H = H; tr->W = W;
    const int rc = unet_run_forward(u, sample, u->tproj, u->tproj_R, out, B, H, W, s, tr);
    if (rc != SISIC_OK) {
        release_tape(u);
        return rc;
    }
    tr->has_tape = true;
    return SISIC_OK;
}

int sisic_unet_backward(sisic_unet* u, const float* dout, void* stream) {
    SISIC_TRY(require_train(u, "backward"));
    TrainState* tr = u->train.get();
    if (!tr->has_tape) {
        set_error("backward: no recorded forward pass (call sisic_unet_train_forward first)");
        return SISIC_ESTATE;
    }
    SISIC_REQUIRE(dout, "backward: null output gradient");
    SISIC_HIP(hipSetDevice(u->ctx->device));
    Bwd b{u, tr, static_cast<hipStream_t>(stream), tr->B};
    const int rc = b.run(dout);
    release_tape(u);       // blocks go back to the pool; the stream still orders later reuse behind the kernels above
    return rc;
}

int sisic_mse_loss(sisic_unet* u, const float* pred, const float* target, int64_t n, float grad_scale, float* loss_dev,
                   float* dpred, void* stream) {
    SISIC_TRY(require_train(u, "mse_loss"));
    return launch_mse(u->ctx, pred, target, (size_t)n, grad_scale, loss_dev ? loss_dev : u->train->loss_dev, dpred,
                      u->train->mse_part, 2048, static_cast<hipStream_t>(stream));
}

int sisic_add_noise(sisic_ctx* ctx, const float* x0, const float* noise, const float* sqrt_alpha_prod,
                    const float* sqrt_one_minus_alpha_prod, float* out, int B, int64_t per_sample, void* stream) {
    SISIC_REQUIRE(ctx && x0 && noise && sqrt_alpha_prod && sqrt_one_minus_alpha_prod && out && B > 0 && per_sample > 0,
                  "add_noise: bad arguments");
    return launch_add_noise(ctx, x0, noise, sqrt_alpha_prod, sqrt_one_minus_alpha_prod, out, B, (size_t)per_sample,
                            static_cast<hipStream_t>(stream));
}

int sisic_unet_optimizer_step(sisic_unet* u, double lr, double beta1, double beta2, double eps, float inv_scale, int* found_inf,
                              void* stream) {
    SISIC_TRY(require_train(u, "optimizer_step"));
    TrainState* tr = u->train.get();
    hipStream_t s = static_cast<hipStream_t>(stream);
    SISIC_HIP(hipSetDevice(u->ctx->device));
    if (found_inf) {                 // GradScaler.step: skip the update when a gradient is inf / nan
        SISIC_HIP(hipMemsetAsync(tr->flag_dev, 0, sizeof(int), s));
        SISIC_TRY(launch_check_finite(u->ctx, tr->grad, u->raw_floats, tr->flag_dev, s));
        int flag = 0;
        SISIC_HIP(hipMemcpyAsync(&flag, tr->flag_dev, sizeof(int), hipMemcpyDeviceToHost, s));
        SISIC_HIP(hipStreamSynchronize(s));
        *found_inf = flag;
        if (flag) return SISIC_OK;
    }
    tr->step += 1;
    SISIC_TRY(launch_adam(u->ctx, u->raw, tr->grad, tr->adam_m, tr->adam_v, u->raw_floats, lr, beta1, beta2, eps, tr->step,
                          inv_scale, s));
    // every derived form of the weights follows the update: three batched launches (repack.hip); the one-launch-per-tensor
    // route of the load path when the buffers have moved since the job tables were built (SISIC_REPACK_BATCH=0: always)
    static const bool batch_on = [] { const char* e = std::getenv("SISIC_REPACK_BATCH"); return !e || std::atoi(e) != 0; }();
    if (batch_on && tr->repack_ready) return run_repack_plan(u, s);
    SISIC_TRY(unet_prepare_all(u, s));
    SISIC_TRY(prepare_backward_weights(u, s));
    if (batch_on) {
        SISIC_HIP(hipStreamSynchronize(s));
        SISIC_TRY(build_repack_plan(u));
    }
    return SISIC_OK;
}

int sisic_unet_train_step(sisic_unet* u, const float* images, const float* noise, const int64_t* timesteps,
                          const float* sqrt_alpha_prod, const float* sqrt_one_minus_alpha_prod, int B, int H, int W, double lr,
                          double beta1, double beta2, double eps, float loss_scale, float* loss_out, int* found_inf,
                          void* stream) {
    SISIC_TRY(require_train(u, "train_step"));
    SISIC_REQUIRE(images && noise && timesteps && sqrt_alpha_prod && sqrt_one_minus_alpha_prod, "train_step: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    TrainState* tr = u->train.get();
    const int C = u->cfg.in_channels;
    const size_t n = (size_t)B * C * H * W;
    SISIC_TRY(unet_grow(&u->eps_buf, &u->eps_floats, 3 * n));     // noisy | prediction | d prediction
    float* noisy = u->eps_buf;
    float* pred = noisy + n;
    float* dpred = pred + n;
    // coefficient rows: host -> device through the pinned ring (2B floats)
    SISIC_TRY(unet_ensure_rows(u, (size_t)B, (size_t)B));
    SISIC_TRY(unet_grow(&tr->small, &tr->small_cap, train_small_floats(u, B)));
    std::vector<float> coef(2 * (size_t)B);
    std::memcpy(coef.data(), sqrt_alpha_prod, B * sizeof(float));
    std::memcpy(coef.data() + B, sqrt_one_minus_alpha_prod, B * sizeof(float));
    SISIC_TRY(unet_stage_upload(u, coef.data(), 2 * (size_t)B, tr->small, s));
    SISIC_TRY(launch_add_noise(u->ctx, images, noise, tr->small, tr->small + B, noisy, B, (size_t)C * H * W, s));
    SISIC_TRY(sisic_unet_train_forward(u, noisy, timesteps, pred, B, H, W, stream));
    SISIC_TRY(launch_mse(u->ctx, pred, noise, n, loss_scale, tr->loss_dev, dpred, tr->mse_part, 2048, s));
    SISIC_TRY(sisic_unet_backward(u, dpred, stream));
    if (loss_out) {
        SISIC_HIP(hipMemcpyAsync(loss_out, tr->loss_dev, sizeof(float), hipMemcpyDeviceToHost, s));
        SISIC_HIP(hipStreamSynchronize(s));
    }
    return sisic_unet_optimizer_step(u, lr, beta1, beta2, eps, 1.0f / loss_scale, found_inf, stream);
}

int sisic_unet_read(sisic_unet* u, int what, int index, float* host_out, int64_t numel) {
    SISIC_REQUIRE(u && host_out && index >= 0 && index < (int)u->names.size(), "unet_read: bad arguments");
    SISIC_REQUIRE(numel == u->numels[index], "unet_read: '%s' has %lld elements", u->names[index].c_str(), (long long)u->numels[index]);
    const float* base = nullptr;
    if (what == 0) base = u->raw;
    else {
        SISIC_TRY(require_train(u, "unet_read"));
        base = what == 1 ? u->train->grad : (what == 2 ? u->train->adam_m : (what == 3 ? u->train->adam_v : nullptr));
    }
    SISIC_REQUIRE(base, "unet_read: what = %d (0 parameter, 1 gradient, 2 Adam m, 3 Adam v)", what);
    SISIC_HIP(hipSetDevice(u->ctx->device));
    SISIC_HIP(hipDeviceSynchronize());
    SISIC_HIP(hipMemcpy(host_out, base + u->offsets[index], (size_t)numel * sizeof(float), hipMemcpyDeviceToHost));
    return SISIC_OK;
}

int64_t sisic_unet_train_steps(const sisic_unet* u) { return (u && u->train) ? u->train->step : 0; }

// ---- single operators of the backward pass (parity-test surface): scratch is allocated per call
int sisic_conv2d_wgrad(sisic_ctx* ctx, const sisic_conv_args* f, const float* dy, float* dw, void* stream) {
    SISIC_REQUIRE(ctx && f && dy && dw, "conv2d_wgrad: null argument");
    SISIC_REQUIRE(f->upsample == 0 || f->upsample == 1, "conv2d_wgrad: upsample mode %d", f->upsample);
    hipStream_t s = static_cast<hipStream_t>(stream);
    WgradArgs a;
    a.in0 = f->in0; a.in1 = f->in1; a.c0 = f->c0; a.c1 = f->c1; a.B = f->B; a.Hin = f->Hin; a.Win = f->Win;
    a.ups = f->upsample; a.ksize = f->ksize; a.stride = f->stride;
    a.gn_scale = f->gn_scale; a.gn_shift = f->gn_shift; a.gn_silu = f->gn_silu;
    a.dy = dy; a.Cout = f->Cout; a.dw = dw;
    const size_t need = conv_wgrad_scratch_floats(a);
    void* part = nullptr;
    SISIC_HIP(hipMalloc(&part, need * sizeof(float)));
    const int rc = launch_conv_wgrad(ctx, a, static_cast<float*>(part), need, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(part);
    return rc;
}

int sisic_attention_bwd(sisic_ctx* ctx, const float* qkv, const float* o, const float* dO, float* dqkv, int B, int C, int N,
                        int head_dim, void* stream) {
    SISIC_REQUIRE(ctx, "attention_bwd: null context");
    return launch_attention_bwd(ctx, qkv, o, dO, dqkv, B, C, N, head_dim, static_cast<hipStream_t>(stream));
}

int sisic_groupnorm_bwd(sisic_ctx* ctx, const float* da, const float* x, int B, int C, int HW, int groups, float eps,
                        const float* gamma, const float* beta, int silu, float* dx_accum, float* dgamma, float* dbeta,
                        void* stream) {
    SISIC_REQUIRE(ctx && da && x && gamma && beta && dx_accum && dgamma && dbeta, "groupnorm_bwd: null argument");
    SISIC_REQUIRE(B > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0, "groupnorm_bwd: bad shape");
    hipStream_t s = static_cast<hipStream_t>(stream);
    void* p = nullptr;
    const size_t n = (size_t)4 * B * C + (size_t)2 * B * groups;
    SISIC_HIP(hipMalloc(&p, n * sizeof(float)));
    float* scale = static_cast<float*>(p);
    float* shift = scale + (size_t)B * C;
    float* sums = shift + (size_t)B * C;
    float* mr = sums + (size_t)2 * B * C;
    int rc = launch_gn_stats(ctx, x, C, nullptr, 0, B, HW, groups, eps, gamma, beta, scale, shift, s, mr);
    if (rc == SISIC_OK)
        rc = launch_gn_bwd(ctx, da, x, C, nullptr, 0, B, HW, groups, scale, shift, mr, gamma, silu, sums, dx_accum, nullptr, dgamma,
                           dbeta, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(p);
    return rc;
}

}  // extern "C"
