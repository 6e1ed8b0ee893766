// resnet.cpp -- forward pass of the ResNet18 classifier used by the reference's explainability passes
// (xai/XAI.py:357-471: torchvision resnet18 with fc -> num_classes, eval mode).
//
// BatchNorm (eval) is folded into the preceding convolution at load time, in float64:
//     w' = w * gamma / sqrt(var + eps),   b' = beta - mean * gamma / sqrt(var + eps)
// so every conv+BN(+ReLU)(+identity) is ONE launch of conv_mfma_kernel with its bias / residual / ReLU
// epilogue.  The pre-processing of XAI.py:399-431 is one fused kernel (classifier.hip).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>

#include "common.h"

using namespace sisic;

namespace {

struct FoldedConv {
    int cout = 0, cin = 0, k = 0, stride = 1;
    int w_idx = -1;
    int bn_w = -1, bn_b = -1, bn_m = -1, bn_v = -1;
    float* packed = nullptr;
    float* wino = nullptr;           // Winograd-domain filters of the BN-folded weight (3x3 stride 1 only)
    float* bias = nullptr;
    // backward-to-input (sisic_resnet_input_gradient): the same convolution with transposed, tap-flipped filters
    float* raw = nullptr;            // BN-folded OIHW weight on the device (the stem's transposed convolution reads it)
    float* packed_t = nullptr;       // packed [cout -> cin] filters W'[ci][co][a][b] = W[co][ci][k-1-a][k-1-b]
    float* wino_t = nullptr;         // their Winograd form (3x3 stride 1 only)
};

struct Block {
    FoldedConv conv1, conv2, down;   // down.k == 0 when the shortcut is the identity
};

struct RBlock {
    float* p;
    size_t bytes;
    bool free_;
};

}  // namespace

struct sisic_resnet {
    sisic_ctx* ctx = nullptr;
    int num_classes = 0;
    std::vector<std::string> names;
    std::vector<int64_t> numels;
    std::map<std::string, int> index;
    std::vector<std::vector<float>> host;   // host copies (folding happens on the host)
    FoldedConv stem;
    std::vector<Block> blocks;
    int fc_w = -1, fc_b = -1;
    float* d_fc_w = nullptr;
    float* d_fc_b = nullptr;
    std::vector<float*> owned;
    std::vector<RBlock> pool;
    int ws_B = 0, ws_H = 0, ws_W = 0;       // shape the pooled blocks were sized for (see workspace_for)
    bool loaded = false;

    int add(const std::string& n, int64_t numel) {
        index[n] = (int)names.size();
        names.push_back(n);
        numels.push_back(numel);
        return (int)names.size() - 1;
    }
};

namespace {

constexpr float BN_EPS = 1e-5f;   // torchvision BatchNorm2d default

void add_conv_bn(sisic_resnet* r, FoldedConv& c, const std::string& conv, const std::string& bn, int cout, int cin, int k,
                 int stride) {
    c.cout = cout; c.cin = cin; c.k = k; c.stride = stride;
    c.w_idx = r->add(conv + ".weight", (int64_t)cout * cin * k * k);
    c.bn_w = r->add(bn + ".weight", cout);
    c.bn_b = r->add(bn + ".bias", cout);
    c.bn_m = r->add(bn + ".running_mean", cout);
    c.bn_v = r->add(bn + ".running_var", cout);
}

void describe(sisic_resnet* r) {
    const std::string pre = "model.";           // XAI.py:389: self.model = models.resnet18(...)
    add_conv_bn(r, r->stem, pre + "conv1", pre + "bn1", 64, 3, 7, 2);
    const int widths[4] = {64, 128, 256, 512};
    int in_ch = 64;
    for (int l = 0; l < 4; ++l) {
        for (int j = 0; j < 2; ++j) {
            Block b;
            const int stride = (l > 0 && j == 0) ? 2 : 1;
            const std::string base = pre + "layer" + std::to_string(l + 1) + "." + std::to_string(j);
            add_conv_bn(r, b.conv1, base + ".conv1", base + ".bn1", widths[l], in_ch, 3, stride);
            add_conv_bn(r, b.conv2, base + ".conv2", base + ".bn2", widths[l], widths[l], 3, 1);
            if (stride != 1 || in_ch != widths[l])
                add_conv_bn(r, b.down, base + ".downsample.0", base + ".downsample.1", widths[l], in_ch, 1, stride);
            r->blocks.push_back(b);
            in_ch = widths[l];
        }
    }
    r->fc_w = r->add(pre + "fc.weight", (int64_t)r->num_classes * 512);
    r->fc_b = r->add(pre + "fc.bias", r->num_classes);
}

int dev_alloc(sisic_resnet* r, size_t floats, float** out) {
    void* p = nullptr;
    SISIC_HIP(hipMalloc(&p, std::max<size_t>(floats, 4) * sizeof(float)));
    r->owned.push_back(static_cast<float*>(p));
    *out = static_cast<float*>(p);
    return SISIC_OK;
}

int fold(sisic_resnet* r, FoldedConv& c) {
    if (c.k == 0) return SISIC_OK;
    const std::vector<float>& w = r->host[c.w_idx];
    const std::vector<float>& g = r->host[c.bn_w];
    const std::vector<float>& be = r->host[c.bn_b];
    const std::vector<float>& m = r->host[c.bn_m];
    const std::vector<float>& v = r->host[c.bn_v];
    const size_t per = (size_t)c.cin * c.k * c.k;
    std::vector<float> wf(w.size()), bf(c.cout);
    for (int co = 0; co < c.cout; ++co) {
        const double sc = (double)g[co] / std::sqrt((double)v[co] + (double)BN_EPS);
        for (size_t i = 0; i < per; ++i) wf[co * per + i] = (float)((double)w[co * per + i] * sc);
        bf[co] = (float)((double)be[co] - (double)m[co] * sc);
    }
    float* raw = nullptr;
    SISIC_TRY(dev_alloc(r, wf.size(), &raw));
    SISIC_HIP(hipMemcpy(raw, wf.data(), wf.size() * sizeof(float), hipMemcpyHostToDevice));
    SISIC_TRY(dev_alloc(r, (size_t)sisic_conv_packed_numel(c.cout, c.cin, c.k), &c.packed));
    SISIC_TRY(launch_conv_pack(r->ctx, raw, c.cout, c.cin, c.k, c.packed, nullptr));
    if (c.k == 3 && c.stride == 1) {     // F(2x2,3x3) for the 13 stride-1 3x3 convolutions (conv_winograd.hip)
        SISIC_TRY(dev_alloc(r, (size_t)winograd_packed_numel(c.cout, c.cin), &c.wino));
        SISIC_TRY(launch_winograd_pack(r->ctx, raw, c.cout, c.cin, c.wino, nullptr));
    }
    SISIC_TRY(dev_alloc(r, c.cout, &c.bias));
    SISIC_HIP(hipMemcpy(c.bias, bf.data(), bf.size() * sizeof(float), hipMemcpyHostToDevice));
    c.raw = raw;
    if (c.k != 7) {                      // backward filters (the 7x7 stem has its own kernel, classifier_bwd.hip)
        const int kk = c.k * c.k;
        std::vector<float> wt(wf.size());
        for (int co = 0; co < c.cout; ++co)
            for (int ci = 0; ci < c.cin; ++ci)
                for (int t = 0; t < kk; ++t)
                    wt[((size_t)ci * c.cout + co) * kk + (kk - 1 - t)] = wf[((size_t)co * c.cin + ci) * kk + t];
        float* rawt = nullptr;
        SISIC_TRY(dev_alloc(r, wt.size(), &rawt));
        SISIC_HIP(hipMemcpy(rawt, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
        SISIC_TRY(dev_alloc(r, (size_t)sisic_conv_packed_numel(c.cin, c.cout, c.k), &c.packed_t));
        SISIC_TRY(launch_conv_pack(r->ctx, rawt, c.cin, c.cout, c.k, c.packed_t, nullptr));
        if (c.k == 3 && c.stride == 1) {
            SISIC_TRY(dev_alloc(r, (size_t)winograd_packed_numel(c.cin, c.cout), &c.wino_t));
            SISIC_TRY(launch_winograd_pack(r->ctx, rawt, c.cin, c.cout, c.wino_t, nullptr));
        }
    }
    return SISIC_OK;
}

int pool_get(sisic_resnet* r, size_t floats, float** out) {
    const size_t bytes = floats * sizeof(float);
    for (auto& b : r->pool)
        if (b.free_ && b.bytes == bytes) { b.free_ = false; *out = b.p; return SISIC_OK; }
    void* p = nullptr;
    SISIC_HIP(hipMalloc(&p, bytes));
    r->pool.push_back({static_cast<float*>(p), bytes, false});
    *out = static_cast<float*>(p);
    return SISIC_OK;
}

void pool_put(sisic_resnet* r, float* p) {
    for (auto& b : r->pool)
        if (b.p == p) { b.free_ = true; return; }
}

// The pool matches blocks by exact size, so every distinct (batch, resolution) would leave its own full set of
// activation buffers behind (the stem output alone is B*64*112*112*4 bytes).  Like the UNet's check_shape, the pool is
// therefore emptied whenever the input shape changes: resident workspace = what the current shape needs, whatever the
// history of batch sizes (trajectory lengths, last chunks of Integrated Gradients, ...) has been.
int workspace_for(sisic_resnet* r, int B, int H, int W) {
    if (r->ws_B == B && r->ws_H == H && r->ws_W == W) return SISIC_OK;
    SISIC_HIP(hipDeviceSynchronize());       // earlier launches may still use the blocks
    for (auto& b : r->pool) (void)hipFree(b.p);
    r->pool.clear();
    r->ws_B = B; r->ws_H = H; r->ws_W = W;
    return SISIC_OK;
}

int run_conv(sisic_resnet* r, const FoldedConv& c, const float* in, int B, int H, int W, const float* residual, bool relu,
             float* out, hipStream_t s) {
    sisic_conv_args a{};
    a.in0 = in; a.c0 = c.cin; a.B = B; a.Hin = H; a.Win = W;
    a.ksize = c.k; a.stride = c.stride;
    a.w_packed = c.packed; a.bias = c.bias; a.Cout = c.cout;
    a.w_winograd = c.wino;
    a.residual = residual; a.relu = relu ? 1 : 0; a.out = out;
    return launch_conv2d(r->ctx, a, s);
}

inline int out_dim(int n, int k, int stride) { return (n + 2 * (k / 2) - k) / stride + 1; }

}  // namespace

extern "C" {

int sisic_resnet_create(sisic_ctx* ctx, int num_classes, sisic_resnet** out) {
    SISIC_REQUIRE(ctx && out && num_classes > 0 && num_classes <= 1000, "resnet_create: bad arguments");
    auto* r = new sisic_resnet();
    r->ctx = ctx;
    r->num_classes = num_classes;
    describe(r);
    *out = r;
    return SISIC_OK;
}

int sisic_resnet_destroy(sisic_resnet* r) {
    if (!r) return SISIC_OK;
    (void)hipDeviceSynchronize();
    for (auto p : r->owned) (void)hipFree(p);
    for (auto& b : r->pool) (void)hipFree(b.p);
    delete r;
    return SISIC_OK;
}

int64_t sisic_resnet_workspace_bytes(const sisic_resnet* r) {
    int64_t n = 0;
    if (r) for (const auto& b : r->pool) n += (int64_t)b.bytes;
    return n;
}

int sisic_resnet_num_tensors(const sisic_resnet* r) { return r ? (int)r->names.size() : 0; }

const char* sisic_resnet_tensor_name(const sisic_resnet* r, int i) {
    if (!r || i < 0 || i >= (int)r->names.size()) return nullptr;
    return r->names[i].c_str();
}

int sisic_resnet_load(sisic_resnet* r, int n, const char* const* names, const float* const* host_ptrs,
                      const int64_t* numels) {
    SISIC_REQUIRE(r && names && host_ptrs && numels, "resnet_load: null argument");
    SISIC_REQUIRE(n == (int)r->names.size(), "resnet_load: state dict has %d float tensors, expected %d", n, (int)r->names.size());
    SISIC_HIP(hipSetDevice(r->ctx->device));
    r->host.assign(r->names.size(), {});
    std::vector<char> seen(r->names.size(), 0);
    for (int i = 0; i < n; ++i) {
        SISIC_REQUIRE(names[i] && host_ptrs[i], "resnet_load: entry %d is null", i);
        auto it = r->index.find(names[i]);
        SISIC_REQUIRE(it != r->index.end(), "resnet_load: unexpected key '%s'", names[i]);
        const int idx = it->second;
        SISIC_REQUIRE(!seen[idx], "resnet_load: duplicate key '%s'", names[i]);
        SISIC_REQUIRE(numels[i] == r->numels[idx], "resnet_load: '%s' has %lld elements, expected %lld", names[i],
                      (long long)numels[i], (long long)r->numels[idx]);
        seen[idx] = 1;
        r->host[idx].assign(host_ptrs[i], host_ptrs[i] + numels[i]);
    }
    (void)hipDeviceSynchronize();
    for (auto p : r->owned) (void)hipFree(p);
    r->owned.clear();
    r->loaded = false;
    SISIC_TRY(fold(r, r->stem));
    for (auto& b : r->blocks) {
        SISIC_TRY(fold(r, b.conv1));
        SISIC_TRY(fold(r, b.conv2));
        SISIC_TRY(fold(r, b.down));
    }
    SISIC_TRY(dev_alloc(r, r->host[r->fc_w].size(), &r->d_fc_w));
    SISIC_TRY(dev_alloc(r, r->host[r->fc_b].size(), &r->d_fc_b));
    SISIC_HIP(hipMemcpy(r->d_fc_w, r->host[r->fc_w].data(), r->host[r->fc_w].size() * sizeof(float), hipMemcpyHostToDevice));
    SISIC_HIP(hipMemcpy(r->d_fc_b, r->host[r->fc_b].data(), r->host[r->fc_b].size() * sizeof(float), hipMemcpyHostToDevice));
    SISIC_HIP(hipDeviceSynchronize());
    r->loaded = true;
    return SISIC_OK;
}

int sisic_resnet_forward(sisic_resnet* r, const float* x, float* logits, int B, int H, int W, int preprocess,
                         void* stream) {
    SISIC_REQUIRE(r && x && logits && B > 0 && H > 0 && W > 0, "resnet_forward: bad arguments");
    if (!r->loaded) {
        set_error("resnet_forward called before sisic_resnet_load");
        return SISIC_ESTATE;
    }
    SISIC_HIP(hipSetDevice(r->ctx->device));
    SISIC_TRY(workspace_for(r, B, H, W));
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<float*> live;
    auto get = [&](size_t floats, float** p) {
        const int rc = pool_get(r, floats, p);
        if (rc == SISIC_OK) live.push_back(*p);
        return rc;
    };
    auto put = [&](float* p) {
        pool_put(r, p);
        live.erase(std::remove(live.begin(), live.end(), p), live.end());
    };
    auto body = [&]() -> int {
        const float* cur = x;
        int h = H, w = W;
        float* pre = nullptr;
        if (preprocess) {
            const int S = 224;                      // CLASSIFIER_IMAGE_SIZE, XAI.py
            SISIC_TRY(get((size_t)B * 3 * S * S, &pre));
            SISIC_TRY(launch_preprocess(r->ctx, x, pre, B, H, W, S, S, s));
            cur = pre; h = S; w = S;
        }
        // stem: conv7x7 s2 (+BN) + ReLU, maxpool 3x3 s2
        int oh = out_dim(h, 7, 2), ow = out_dim(w, 7, 2);
        float* c1 = nullptr;
        SISIC_TRY(get((size_t)B * 64 * oh * ow, &c1));
        SISIC_TRY(run_conv(r, r->stem, cur, B, h, w, nullptr, true, c1, s));
        if (pre) put(pre);
        h = oh; w = ow;
        oh = out_dim(h, 3, 2); ow = out_dim(w, 3, 2);
        float* act = nullptr;
        SISIC_TRY(get((size_t)B * 64 * oh * ow, &act));
        SISIC_TRY(launch_maxpool(r->ctx, c1, act, B, 64, h, w, s));
        put(c1);
        h = oh; w = ow;
        int ch = 64;
        for (const Block& b : r->blocks) {
            const int bh = out_dim(h, 3, b.conv1.stride), bw = out_dim(w, 3, b.conv1.stride);
            float* t1 = nullptr;
            SISIC_TRY(get((size_t)B * b.conv1.cout * bh * bw, &t1));
            SISIC_TRY(run_conv(r, b.conv1, act, B, h, w, nullptr, true, t1, s));
            const float* identity = act;
            float* ds = nullptr;
            if (b.down.k) {
                SISIC_TRY(get((size_t)B * b.down.cout * bh * bw, &ds));
                SISIC_TRY(run_conv(r, b.down, act, B, h, w, nullptr, false, ds, s));
                identity = ds;
            }
            float* t2 = nullptr;
            SISIC_TRY(get((size_t)B * b.conv2.cout * bh * bw, &t2));
            SISIC_TRY(run_conv(r, b.conv2, t1, B, bh, bw, identity, true, t2, s));    // relu(bn2(conv2) + identity)
            put(t1);
            if (ds) put(ds);
            put(act);
            act = t2; h = bh; w = bw; ch = b.conv2.cout;
        }
        SISIC_TRY(launch_avgpool_fc(r->ctx, act, r->d_fc_w, r->d_fc_b, logits, B, ch, h * w, r->num_classes, s));
        put(act);
        return SISIC_OK;
    };
    const int rc = body();
    for (float* p : live) pool_put(r, p);
    return rc;
}

// The stem's activation relu(bn1(conv1(pre(x)))) alone: what the 3x3/2 max-pool chooses its arg-maxima from.  Parity tests
// replay these routes in the CPU autograd pass (tests/test_gpu_classifier.py), so that input-gradient parity is a
// max-abs statement instead of a statistical one.
int sisic_resnet_stem(sisic_resnet* r, const float* x, float* c1_out, int B, int H, int W, int preprocess, void* stream) {
    SISIC_REQUIRE(r && x && c1_out && B > 0 && H > 0 && W > 0, "resnet_stem: bad arguments");
    if (!r->loaded) {
        set_error("resnet_stem called before sisic_resnet_load");
        return SISIC_ESTATE;
    }
    SISIC_HIP(hipSetDevice(r->ctx->device));
    SISIC_TRY(workspace_for(r, B, H, W));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float* cur = x;
    int h = H, w = W;
    float* pre = nullptr;
    if (preprocess) {
        const int S = 224;
        SISIC_TRY(pool_get(r, (size_t)B * 3 * S * S, &pre));
        const int rc = launch_preprocess(r->ctx, x, pre, B, H, W, S, S, s);
        if (rc != SISIC_OK) { pool_put(r, pre); return rc; }
        cur = pre; h = S; w = S;
    }
    const int rc = run_conv(r, r->stem, cur, B, h, w, nullptr, true, c1_out, s);
    if (pre) pool_put(r, pre);
    return rc;
}

// d score / d x for score = log(softmax(logits)[target] + 1e-8) (XAI.py:443-459), x the classifier's raw input in
// [-1,1] (pre-processing included): the quantity captum's IntegratedGradients and the plain-gradient fallback of
// XAI.py:1039-1109 differentiate.  Forward with the activations kept, then the transposed network (see
// classifier_bwd.hip); every convolution of the backward pass is a sisic_conv2d launch with transposed filters.
int sisic_resnet_input_gradient(sisic_resnet* r, const float* x, int B, int H, int W, int target, float* grad_x,
                                float* logits_out, void* stream) {
    SISIC_REQUIRE(r && x && grad_x && B > 0 && H > 0 && W > 0, "resnet_input_gradient: bad arguments");
    SISIC_REQUIRE(target >= 0 && target < r->num_classes, "resnet_input_gradient: class %d of %d", target, r->num_classes);
    if (!r->loaded) {
        set_error("resnet_input_gradient called before sisic_resnet_load");
        return SISIC_ESTATE;
    }
    SISIC_HIP(hipSetDevice(r->ctx->device));
    SISIC_TRY(workspace_for(r, B, H, W));
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<float*> live;
    auto get = [&](size_t floats, float** p) {
        const int rc = pool_get(r, floats, p);
        if (rc == SISIC_OK) live.push_back(*p);
        return rc;
    };
    auto put = [&](float* p) {
        pool_put(r, p);
        live.erase(std::remove(live.begin(), live.end(), p), live.end());
    };
    // transposed convolution of `c` applied to g [B, c.cout, gh, gw]; stride 2: zero-insertion input (2gh x 2gw grid)
    auto conv_t = [&](const FoldedConv& c, const float* g, int gh, int gw, const float* residual, float* out) {
        sisic_conv_args a{};
        a.in0 = g; a.c0 = c.cout; a.B = B; a.Hin = gh; a.Win = gw;
        a.ksize = c.k; a.stride = 1;
        a.upsample = (c.k == 3 && c.stride == 2) ? 2 : 0;
        a.w_packed = c.packed_t; a.w_winograd = c.wino_t; a.Cout = c.cin;
        a.residual = residual; a.out = out;
        return launch_conv2d(r->ctx, a, s);
    };
    struct Saved { float* t1; float* out; int h, w, bh, bw; };
    auto body = [&]() -> int {
        const int S = 224;
        SISIC_REQUIRE(H <= S && W <= S, "resnet_input_gradient: input %dx%d larger than the classifier's %dx%d", H, W, S, S);
        float* pre = nullptr;
        SISIC_TRY(get((size_t)B * 3 * S * S, &pre));
        SISIC_TRY(launch_preprocess(r->ctx, x, pre, B, H, W, S, S, s));
        int h = S, w = S;
        const int c1h = out_dim(h, 7, 2), c1w = out_dim(w, 7, 2);
        float* c1 = nullptr;
        SISIC_TRY(get((size_t)B * 64 * c1h * c1w, &c1));
        SISIC_TRY(run_conv(r, r->stem, pre, B, h, w, nullptr, true, c1, s));
        put(pre);
        const int mh = out_dim(c1h, 3, 2), mw = out_dim(c1w, 3, 2);
        SISIC_REQUIRE(c1h % 2 == 0 && c1w % 2 == 0, "resnet_input_gradient: odd feature map");
        float* m = nullptr;
        SISIC_TRY(get((size_t)B * 64 * mh * mw, &m));
        SISIC_TRY(launch_maxpool(r->ctx, c1, m, B, 64, c1h, c1w, s));
        std::vector<Saved> saved;
        float* act = m;
        h = mh; w = mw;
        for (const Block& b : r->blocks) {
            const int bh = out_dim(h, 3, b.conv1.stride), bw = out_dim(w, 3, b.conv1.stride);
            SISIC_REQUIRE(b.conv1.stride == 1 || (h % 2 == 0 && w % 2 == 0), "resnet_input_gradient: odd feature map %dx%d", h, w);
            float* t1 = nullptr;
            SISIC_TRY(get((size_t)B * b.conv1.cout * bh * bw, &t1));
            SISIC_TRY(run_conv(r, b.conv1, act, B, h, w, nullptr, true, t1, s));
            const float* identity = act;
            float* ds = nullptr;
            if (b.down.k) {
                SISIC_TRY(get((size_t)B * b.down.cout * bh * bw, &ds));
                SISIC_TRY(run_conv(r, b.down, act, B, h, w, nullptr, false, ds, s));
                identity = ds;
            }
            float* t2 = nullptr;
            SISIC_TRY(get((size_t)B * b.conv2.cout * bh * bw, &t2));
            SISIC_TRY(run_conv(r, b.conv2, t1, B, bh, bw, identity, true, t2, s));
            if (ds) put(ds);
            saved.push_back({t1, t2, h, w, bh, bw});
            act = t2; h = bh; w = bw;
        }
        const int C = r->blocks.back().conv2.cout;
        float* logits = nullptr;
        SISIC_TRY(get((size_t)B * r->num_classes, &logits));
        SISIC_TRY(launch_avgpool_fc(r->ctx, act, r->d_fc_w, r->d_fc_b, logits, B, C, h * w, r->num_classes, s));
        if (logits_out)
            SISIC_HIP(hipMemcpyAsync(logits_out, logits, (size_t)B * r->num_classes * sizeof(float), hipMemcpyDeviceToDevice, s));

        // ---- backward
        float* g = nullptr;                                   // d score / d (block output), already ReLU-masked
        SISIC_TRY(get((size_t)B * C * h * w, &g));
        SISIC_TRY(launch_score_head_bwd(r->ctx, logits, r->d_fc_w, act, g, B, C, h * w, r->num_classes, target, s));
        put(logits);
        for (int k = (int)r->blocks.size() - 1; k >= 0; --k) {
            const Block& b = r->blocks[k];
            const Saved& sv = saved[k];
            const size_t n_mid = (size_t)B * b.conv2.cout * sv.bh * sv.bw;
            float* tmp = nullptr;                             // conv2^T g, then the ReLU mask of t1
            SISIC_TRY(get(n_mid, &tmp));
            SISIC_TRY(conv_t(b.conv2, g, sv.bh, sv.bw, nullptr, tmp));
            SISIC_TRY(launch_relu_bwd(r->ctx, tmp, sv.t1, tmp, (int64_t)n_mid, s));
            const size_t n_in = (size_t)B * b.conv1.cin * sv.h * sv.w;
            float* da = nullptr;
            SISIC_TRY(get(n_in, &da));
            if (!b.down.k) {
                SISIC_TRY(conv_t(b.conv1, tmp, sv.bh, sv.bw, g, da));                 // + identity path
            } else {
                SISIC_TRY(conv_t(b.conv1, tmp, sv.bh, sv.bw, nullptr, da));           // stride 2: zero-insertion input
                float* small = nullptr;
                SISIC_TRY(get((size_t)B * b.down.cin * sv.bh * sv.bw, &small));
                SISIC_TRY(conv_t(b.down, g, sv.bh, sv.bw, nullptr, small));           // 1x1 at the low resolution
                SISIC_TRY(launch_scatter_add_even(r->ctx, da, small, B * b.down.cin, sv.h, sv.w, s));
                put(small);
            }
            put(tmp);
            put(g);
            put(sv.t1);
            put(sv.out);
            if (k > 0) {                                       // ReLU of the previous block's output
                SISIC_TRY(launch_relu_bwd(r->ctx, da, saved[k - 1].out, da, (int64_t)n_in, s));
            }
            g = da;
        }
        // g = d score / d (max-pool output)
        float* dc1 = nullptr;
        SISIC_TRY(get((size_t)B * 64 * c1h * c1w, &dc1));
        SISIC_TRY(launch_maxpool_bwd(r->ctx, g, c1, dc1, B * 64, c1h, c1w, s));       // includes the stem's ReLU mask
        put(g);
        put(m);
        put(c1);
        float* dp = nullptr;
        SISIC_TRY(get((size_t)B * 3 * S * S, &dp));
        SISIC_TRY(launch_stem_bwd(r->ctx, dc1, r->stem.raw, dp, B, 64, c1h, c1w, S, S, s));
        put(dc1);
        SISIC_TRY(launch_preprocess_bwd(r->ctx, dp, x, grad_x, B, H, W, S, S, s));
        put(dp);
        return SISIC_OK;
    };
    const int rc = body();
    for (float* p : live) pool_put(r, p);
    return rc;
}

// Grad-CAM on layer4[-1].conv2 for the raw logit of `target` (xai/XAI.py:2945-3035; see gradcam_kernel): the forward
// pass with the last block's conv2 (+BatchNorm) output kept before the residual add, then one small kernel per image.
// cam: dev [B,224,224] in [0,1] (pytorch_grad_cam's double min-max scaling), logits_out: dev [B,n_classes] or NULL.
int sisic_resnet_gradcam(sisic_resnet* r, const float* x, int B, int H, int W, int target, float* cam, float* logits_out,
                         void* stream) {
    SISIC_REQUIRE(r && x && cam && B > 0 && H > 0 && W > 0, "resnet_gradcam: bad arguments");
    SISIC_REQUIRE(target >= 0 && target < r->num_classes, "resnet_gradcam: class %d of %d", target, r->num_classes);
    if (!r->loaded) {
        set_error("resnet_gradcam called before sisic_resnet_load");
        return SISIC_ESTATE;
    }
    SISIC_HIP(hipSetDevice(r->ctx->device));
    SISIC_TRY(workspace_for(r, B, H, W));
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<float*> live;
    auto get = [&](size_t floats, float** p) {
        const int rc = pool_get(r, floats, p);
        if (rc == SISIC_OK) live.push_back(*p);
        return rc;
    };
    auto put = [&](float* p) {
        pool_put(r, p);
        live.erase(std::remove(live.begin(), live.end(), p), live.end());
    };
    auto body = [&]() -> int {
        const int S = 224;
        SISIC_REQUIRE(H <= S && W <= S, "resnet_gradcam: input %dx%d larger than the classifier's %dx%d", H, W, S, S);
        float* pre = nullptr;
        SISIC_TRY(get((size_t)B * 3 * S * S, &pre));
        SISIC_TRY(launch_preprocess(r->ctx, x, pre, B, H, W, S, S, s));
        int h = S, w = S;
        int oh = out_dim(h, 7, 2), ow = out_dim(w, 7, 2);
        float* c1 = nullptr;
        SISIC_TRY(get((size_t)B * 64 * oh * ow, &c1));
        SISIC_TRY(run_conv(r, r->stem, pre, B, h, w, nullptr, true, c1, s));
        put(pre);
        h = oh; w = ow;
        oh = out_dim(h, 3, 2); ow = out_dim(w, 3, 2);
        float* act = nullptr;
        SISIC_TRY(get((size_t)B * 64 * oh * ow, &act));
        SISIC_TRY(launch_maxpool(r->ctx, c1, act, B, 64, h, w, s));
        put(c1);
        h = oh; w = ow;
        float* ylast = nullptr;
        for (size_t k = 0; k < r->blocks.size(); ++k) {
            const Block& b = r->blocks[k];
            const bool last = k + 1 == r->blocks.size();
            const int bh = out_dim(h, 3, b.conv1.stride), bw = out_dim(w, 3, b.conv1.stride);
            float* t1 = nullptr;
            SISIC_TRY(get((size_t)B * b.conv1.cout * bh * bw, &t1));
            SISIC_TRY(run_conv(r, b.conv1, act, B, h, w, nullptr, true, t1, s));
            const float* identity = act;
            float* ds = nullptr;
            if (b.down.k) {
                SISIC_TRY(get((size_t)B * b.down.cout * bh * bw, &ds));
                SISIC_TRY(run_conv(r, b.down, act, B, h, w, nullptr, false, ds, s));
                identity = ds;
            }
            float* t2 = nullptr;
            const size_t n_out = (size_t)B * b.conv2.cout * bh * bw;
            SISIC_TRY(get(n_out, &t2));
            if (!last) {
                SISIC_TRY(run_conv(r, b.conv2, t1, B, bh, bw, identity, true, t2, s));
            } else {                                           // keep bn2(conv2(.)) on its own, then relu(. + identity)
                SISIC_TRY(get(n_out, &ylast));
                SISIC_TRY(run_conv(r, b.conv2, t1, B, bh, bw, nullptr, false, ylast, s));
                SISIC_TRY(launch_add_relu(r->ctx, ylast, identity, t2, (int64_t)n_out, s));
            }
            put(t1);
            if (ds) put(ds);
            put(act);
            act = t2; h = bh; w = bw;
        }
        const FoldedConv& c2 = r->blocks.back().conv2;
        float* logits = nullptr;
        SISIC_TRY(get((size_t)B * r->num_classes, &logits));
        SISIC_TRY(launch_avgpool_fc(r->ctx, act, r->d_fc_w, r->d_fc_b, logits, B, c2.cout, h * w, r->num_classes, s));
        if (logits_out)
            SISIC_HIP(hipMemcpyAsync(logits_out, logits, (size_t)B * r->num_classes * sizeof(float), hipMemcpyDeviceToDevice, s));
        put(logits);
        SISIC_TRY(launch_gradcam(r->ctx, ylast, act, r->d_fc_w, c2.bias, cam, B, c2.cout, h, w, S, target, s));
        put(ylast);
        put(act);
        return SISIC_OK;
    };
    const int rc = body();
    for (float* p : live) pool_put(r, p);
    return rc;
}

int sisic_class_scores(sisic_ctx* ctx, const float* logits, int B, int n_classes, int target, float* prob,
                       float* logscore, void* stream) {
    SISIC_REQUIRE(ctx, "class_scores: null context");
    return launch_class_scores(ctx, logits, B, n_classes, target, prob, logscore, static_cast<hipStream_t>(stream));
}

int sisic_mask_patches(sisic_ctx* ctx, const float* image, const uint8_t* masks, float* out, int S, int C, int H, int W,
                       int patch, void* stream) {
    SISIC_REQUIRE(ctx, "mask_patches: null context");
    return launch_mask_patches(ctx, image, masks, out, S, C, H, W, patch, static_cast<hipStream_t>(stream));
}

}  // extern "C"
