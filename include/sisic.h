/*
 * sisic.h -- C ABI of libsisic_hip.so, the MI355X (gfx950) implementation of the
 * SYNT_ISIC DDPM sampling hot path.
 *
 * The reference (fims9000/SYNT_ISIC) is pure Python: it has no FFI, the boundary
 * of its hot path is duck-typing on two third-party objects
 *     noise_pred = model(latents, t).sample                         core/generator/image_generator.py:400
 *     latents    = scheduler.step(noise_pred, t, latents).prev_sample               image_generator.py:403
 * built at core/generator/model_manager.py:173-194 (UNet2DModel) and :196-212
 * (DDPMScheduler).  A maintainer of the reference binds this library with
 * ctypes; synt_isic_amd/_lib.py is that binding.
 *
 * Conventions
 *   - every function returns 0 (SISIC_OK) or a negative SISIC_E* code; the message
 *     of the last failure on the calling thread is sisic_last_error();
 *   - "dev" pointers are device (HBM) addresses of contiguous fp32 NCHW tensors
 *     owned by the caller (torch); the library never frees or reallocates them;
 *   - weights are copied into a library-owned arena at load time; workspace is
 *     library-owned per handle and grows on demand (never inside the sampling loop);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     all work is enqueued asynchronously on it unless stated otherwise;
 *   - handles are not thread-safe; use one handle per (device, stream).
 */
#ifndef SISIC_H
#define SISIC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): sisic_sample_frames added; the packed-filter buffers grew in round 3 (1x1: three layouts = 3.5 x the
 * [Cin_pad][1][Cout_pad] floats; Winograd: the f32 U plus the bf16x3 split of it) -- ALWAYS size them with the *_numel
 * functions below, never from the layout comment. */
#define SISIC_ABI_VERSION 3

#define SISIC_OK 0
#define SISIC_EINVAL (-1)   /* bad argument / unsupported shape */
#define SISIC_EHIP (-2)     /* a HIP runtime call failed        */
#define SISIC_ESTATE (-3)   /* call order violated (e.g. forward before load) */
#define SISIC_ECANCEL (-4)  /* sampling loop stopped by the cancel flag */

typedef struct sisic_ctx sisic_ctx;     /* device context: weights-independent scratch */
typedef struct sisic_unet sisic_unet;   /* one UNet2DModel instance (weights + workspace) */
typedef struct sisic_resnet sisic_resnet; /* one ResNet18 classifier instance */

int sisic_abi_version(void);
const char* sisic_last_error(void);

/* ---- context ------------------------------------------------------------------ */
int sisic_create(int device_id, sisic_ctx** out);
int sisic_destroy(sisic_ctx* ctx);

/* ---- single operators (parity-test surface; also what the model executor calls) -- */

/* Generic NCHW fp32 convolution on the f32 MFMA pipe, replaces the F.conv2d /
 * GroupNorm / SiLU / cat / interpolate instances inside diffusers' ResnetBlock2D,
 * Downsample2D, Upsample2D and Attention projections.
 *   out[b,co,y,x] = bias[co] + chan_bias[b,co] + residual[b,co,y,x]
 *                 + sum_{ci,ky,kx} W[co,ci,ky,kx] * act(in[b,ci,...])
 * where `in` is the channel concatenation of in0 (c0 channels) and in1 (c1 channels,
 * may be NULL), optionally nearest-upsampled 2x, and act(v) is the optional
 * GroupNorm-apply prologue v*gn_scale[b,ci]+gn_shift[b,ci] followed by SiLU when
 * gn_silu != 0.  Zero padding ksize/2 is applied AFTER the prologue.            */
typedef struct sisic_conv_args {
    const float* in0;       /* dev [B,c0,Hin,Win]                        */
    const float* in1;       /* dev [B,c1,Hin,Win] or NULL                */
    int c0, c1;
    int B, Hin, Win;
    int upsample;           /* 1: nearest 2x before the conv; 2: zero insertion (source at the even
                               coordinates of the 2x grid, zeros elsewhere) = the input side of a
                               transposed stride-2 convolution (classifier backward-to-input)        */
    int ksize;              /* 1, 3 or 7                                  */
    int stride;             /* 1 or 2                                     */
    const float* w_packed;  /* dev, layout of sisic_conv_pack_weights     */
    const float* bias;      /* dev [Cout] or NULL                         */
    int Cout;
    const float* gn_scale;  /* dev [B,c0+c1] or NULL (no prologue)        */
    const float* gn_shift;  /* dev [B,c0+c1] or NULL                      */
    int gn_silu;
    const float* chan_bias; /* dev [B,Cout] or NULL                       */
    int chan_bias_stride;   /* floats between samples of chan_bias; 0 = one row shared by all samples */
    const float* residual;  /* dev [B,Cout,Hout,Wout] or NULL; MAY BE `out` itself (in-place accumulate, out += conv(...)):
                             * every kernel reads a residual element in the thread that stores that element, before the
                             * store (tests/test_gpu_kernels.py::test_conv_residual_may_alias_out).  No other argument
                             * may overlap `out`.                         */
    int relu;               /* 1: max(0, .) after everything (classifier) */
    float* out;             /* dev [B,Cout,Hout,Wout]                     */
    int tile_cfg;           /* 0 = auto; >0 forces a tile configuration (tests/tuning) */
    const float* w_winograd; /* dev, layout of sisic_conv_winograd_pack, or NULL: when given, 3x3 stride-1
                                convolutions may run as Winograd F(2x2,3x3) (tile_cfg 60..74 force a form; 74 = the
                                fp32-equivalent bf16x3 form, the automatic choice for whole 64-channel x 16x16-pixel
                                tiles; 1x1: tile_cfg 28 = the bf16x3 pointwise kernel, 20 = the f32 one)             */
    float* stats_out;        /* dev [B,Cout,slots,4] or NULL: each workgroup also writes (count, sum, sum of squared
                                deviations from its own mean, 0) of the values it stored, per image and channel, so that the GroupNorm
                                that follows needs sisic_groupnorm_finalize only (no second pass over `out`).
                                slots = sisic_conv_stats_slots(args); 0 there means "not available for this
                                launch" and stats_out must stay NULL                                          */
    /* Optional (ABI 3): the launch also FINALIZES the GroupNorm that reads its output alone -- groups over Cout, the module's
     * gamma / beta -- and leaves that layer's (scale, shift) [B,Cout] (what sisic_groupnorm_finalize would compute from
     * stats_out, bit for bit), so that no finalisation launch follows.  Only where sisic_conv_finalizes(args) says so
     * (today: the K-split 8x8-level forms with 8 channels per group); elsewhere the fields are ignored.                */
    const float* fin_gamma;  /* dev [Cout] or NULL (NULL: none of this)   */
    const float* fin_beta;   /* dev [Cout]                                */
    int fin_groups;
    float fin_eps;
    float* fin_scale;        /* dev [B,Cout] out                          */
    float* fin_shift;        /* dev [B,Cout] out                          */
    float* fin_mean_rstd;    /* dev [B,groups,2] out or NULL              */
} sisic_conv_args;

/* 1 when sisic_conv2d(args) with fin_gamma set will write fin_scale / fin_shift, 0 when the kernel selected for these
 * arguments does not (the caller then runs sisic_groupnorm_finalize on stats_out as before).                          */
int sisic_conv_finalizes(const sisic_conv_args* args);

/* Number of partial-statistics slots per (image, output channel) that sisic_conv2d(args) writes to
 * args->stats_out, or 0 when the kernel selected for these arguments does not produce them.      */
int sisic_conv_stats_slots(const sisic_conv_args* args);

/* Winograd-domain filters U = G g G^T of an OIHW 3x3 weight, computed in float64:
 * number of floats, and dev OIHW -> dev packed [Cin_pad][16][Cout_pad], FOLLOWED by the same filters split into three bf16
 * terms in the operand order of the bf16x3 kernels (pack_device.h).  The buffer must hold sisic_conv_winograd_numel floats:
 * more than the f32 layout alone.                                                             */
int64_t sisic_conv_winograd_numel(int Cout, int Cin);
int sisic_conv_winograd_pack(sisic_ctx*, const float* w_oihw, int Cout, int Cin, float* u_packed, void* stream);

/* number of floats of the packed form of an OIHW weight [Cout,Cin,k,k] (-1: unsupported ksize).  For ksize 1 this is 3.5 x
 * the [Cin_pad][1][Cout_pad] layout: the direct kernel's, the pointwise kernel's and the bf16x3 split are all written. */
int64_t sisic_conv_packed_numel(int Cout, int Cin, int ksize);
/* dev OIHW -> dev packed [Cin_pad][k*k][Cout_pad] (+ the further layouts above), zero padded; w_packed must hold
 * sisic_conv_packed_numel floats */
int sisic_conv_pack_weights(sisic_ctx*, const float* w_oihw, int Cout, int Cin, int ksize,
                            float* w_packed, void* stream);
int sisic_conv2d(sisic_ctx*, const sisic_conv_args* args, void* stream);

/* GroupNorm statistics folded with the affine parameters (replaces the reduction
 * half of torch.nn.GroupNorm inside ResnetBlock2D.norm1/norm2, Attention.group_norm
 * and conv_norm_out):  for c in group g of sample b
 *     scale[b,c] = gamma[c]*rstd[b,g],  shift[b,c] = beta[c] - mean[b,g]*scale[b,c].
 * The input is the concatenation of in0/in1 as in sisic_conv_args; HW = H*W.      */
int sisic_groupnorm_stats(sisic_ctx*, const float* in0, int c0, const float* in1, int c1,
                          int B, int HW, int groups, float eps,
                          const float* gamma, const float* beta,
                          float* scale, float* shift, void* stream);

/* GroupNorm statistics from convolution-epilogue partials (sisic_conv_args.stats_out) instead of a pass
 * over the tensor: same outputs as sisic_groupnorm_stats for the concatenation of two producers'
 * outputs (stats1 NULL / c1 = 0 for one).  The partials are merged in float64 (sum of M2_i + n_i (mean_i - mean)^2) in a fixed order.
 * HW is accepted for symmetry with sisic_groupnorm_stats; the element counts travel with the partials.   */
int sisic_groupnorm_finalize(sisic_ctx*, const float* stats0, int c0, int slots0,
                             const float* stats1, int c1, int slots1,
                             int B, int HW, int groups, float eps,
                             const float* gamma, const float* beta,
                             float* scale, float* shift, void* stream);

/* Multi-head self-attention core (replaces scaled_dot_product_attention inside
 * diffusers' Attention, heads = C/head_dim, softmax in fp32, scale head_dim^-0.5).
 * qkv: dev [B,3*C,N] (q channels, then k, then v; channel = head*head_dim + d),
 * out: dev [B,C,N].  head_dim must be 8 (the reference's attention_head_dim).     */
int sisic_attention(sisic_ctx*, const float* qkv, float* out, int B, int C, int N, int head_dim,
                    void* stream);

/* Fused DDPMScheduler.step, elementwise over n floats:
 *   x0 = clamp((x - sqrt_beta_prod*eps)/sqrt_alpha_prod, -clip, clip)   (clip<=0: no clamp)
 *   out = (c0*x0 + c1*x) + sigma*z                                      (z==NULL or sigma==0: no noise)
 * evaluated in exactly that fp32 operation order (no FMA contraction).            */
int sisic_ddpm_step(sisic_ctx*, const float* eps, const float* x, const float* z, float* out,
                    int64_t n, float sqrt_beta_prod, float sqrt_alpha_prod, float c0, float c1,
                    float sigma, float clip, void* stream);

/* De-normalise image_generator.py:441-447: [B,3,H,W] fp32 -> uint8 [B,H,W,3],
 * trunc(clamp((x+1)/2,0,1)*255).                                                   */
int sisic_denorm_u8(sisic_ctx*, const float* x, uint8_t* out, int B, int C, int H, int W, void* stream);
/* The reference's three spellings of the same conversion, each in its own fp32 operation order:
 *   form 0  image_generator.py:441-447       trunc(clamp((x+1)/2, 0, 1) * 255)            (= sisic_denorm_u8)
 *   form 1  generate_test.py:94-97           trunc((clamp(x,-1,1) + 1) * 0.5 * 255)       (bit-equal to form 0)
 *   form 2  diffusion_generator.py:231-232   trunc(clip((x+1) * 127.5, 0, 255))           (bit-equal to form 0 as well:
 *           halving is exact in binary floating point, so (x+1)/2*255 and (x+1)*127.5 round the same real number once;
 *           tests/test_gpu_kernels.py::test_denorm_u8_three_reference_forms_bit_exact asserts it on every boundary)    */
int sisic_denorm_u8_form(sisic_ctx*, const float* x, uint8_t* out, int B, int C, int H, int W, int form, void* stream);

/* ---- UNet2DModel ---------------------------------------------------------------- */
typedef struct sisic_unet_config {
    int in_channels, out_channels;
    int layers_per_block;
    int n_blocks;                    /* <= 8 */
    int block_out_channels[8];
    int down_attn[8];                /* 1 = AttnDownBlock2D */
    int up_attn[8];                  /* 1 = AttnUpBlock2D   */
    int norm_groups;
    float norm_eps;
    int head_dim;
    int n_freqs;                     /* block_out_channels[0] / 2 */
    const float* freqs;              /* host [n_freqs]: sinusoid frequencies (fp32, copied) */
} sisic_unet_config;

int sisic_unet_create(sisic_ctx*, const sisic_unet_config* cfg, sisic_unet** out);
int sisic_unet_destroy(sisic_unet*);
/* number / names of the tensors load expects (diffusers state_dict key order) */
int sisic_unet_num_tensors(const sisic_unet*);
const char* sisic_unet_tensor_name(const sisic_unet*, int index);
/* Strict load of a flat state dict: n host pointers to contiguous fp32 tensors with
 * the given element counts; every expected tensor must be present exactly once.    */
int sisic_unet_load(sisic_unet*, int n, const char* const* names, const float* const* host_ptrs,
                    const int64_t* numels);
/* Latency mode (off by default): kernel choices that let ONE image fill the chip -- the reference generates its images one
 * at a time (image_generator.py:379, batch 1) -- at the price of extra partial-sum traffic that costs throughput at large
 * batches.  The choice depends on the layer shapes only, so results stay independent of the batch WITHIN a mode; between
 * the two modes results differ in the last bits (different summation order over the input channels).                  */
int sisic_unet_set_latency_mode(sisic_unet*, int on);
/* sisic_sample as one captured step (hipGraph) replayed T-1 times instead of ~190 launches per step from the host:
 * mode 1 on, 0 off, -1 (default) on exactly when latency mode is on.  Same kernels, same arithmetic, same bits; the
 * loop then runs on a library-owned copy of x (written back at the end) so that every address in the graph is stable.   */
int sisic_unet_set_graph_mode(sisic_unet*, int mode);
/* How many times this handle has captured + instantiated the sampling step (a second sisic_sample at the same shape,
 * stream and mode replays the cached graph: the count does not move).                                                  */
int64_t sisic_unet_graph_builds(const sisic_unet*);
/* eps = model(sample, timestep).sample.  timesteps: host int64 [B] (one per sample). */
int sisic_unet_forward(sisic_unet*, const float* sample, const int64_t* timesteps,
                       float* out, int B, int H, int W, void* stream);

/* The whole reverse-diffusion loop (image_generator.py:395-403) on one stream:
 *   for i in 0..T-1:  eps = unet(x, t[i]);  x = ddpm_step(eps, x, z[i], coef[i])
 * x: dev [B,C,H,W], updated in place (x_T in, x_0 out).
 * timesteps: host int64 [T].  coef: host float [T*5] = per step
 *   {sqrt_beta_prod, sqrt_alpha_prod, c0, c1, sigma}.
 * noise: dev [n_noise,B,C,H,W] consumed in order by the steps with sigma != 0, or NULL.
 * traj: dev [T,B,C,H,W] receiving x after every step, or NULL.
 * out_u8: dev uint8 [B,H,W,C] final de-normalised image, or NULL.
 * cancel: host int* polled between steps (non-zero stops the loop with SISIC_ECANCEL), or NULL.
 * steps_done: host int* receiving the number of completed steps, or NULL.           */
int sisic_sample(sisic_unet*, float* x, int B, int H, int W, int T, const int64_t* timesteps,
                 const float* coef, float clip, const float* noise, float* traj, uint8_t* out_u8,
                 const volatile int* cancel, int* steps_done, void* stream);
/* The same loop keeping only SOME frames of the trajectory (xai/XAI.py:751-757 `save_indices`, :815-826: every N-th step and
 * the last one, or the steps whose t is a multiple of N): traj_row is a host int [T], traj_row[i] = the row of traj that
 * receives x after step i, or -1 for a step that is not kept (traj must hold max(traj_row)+1 rows of B*C*H*W floats).
 * traj_row NULL keeps every step in row i (= sisic_sample).                                                              */
int sisic_sample_frames(sisic_unet*, float* x, int B, int H, int W, int T, const int64_t* timesteps,
                        const float* coef, float clip, const float* noise, float* traj, const int* traj_row,
                        uint8_t* out_u8, const volatile int* cancel, int* steps_done, void* stream);

/* ---- training step (diffusion/train_diffusion.py:201-266) -----------------------------------
 * fp32 throughout.  The reference wraps the forward in torch.cuda.amp.autocast() (fp16 matmuls/convolutions) and scales the
 * loss with a GradScaler; here the GradScaler PROTOCOL is implemented (loss_scale multiplies d loss, the optimizer step
 * unscales, checks for inf/nan and skips), the arithmetic stays fp32.
 *
 * train_begin allocates the gradient and Adam (m, v) arenas (zeroed, step count 0) and the filters of the backward-data
 * convolutions; train_end frees them.  One tape at a time: train_forward records it, backward consumes it.           */
int sisic_unet_train_begin(sisic_unet*);
int sisic_unet_train_end(sisic_unet*);
/* DDPMScheduler.add_noise (train_diffusion.py:217): out[b] = sqrt_alpha_prod[b] * x0[b] + sqrt_one_minus_alpha_prod[b] *
 * noise[b]; the two coefficient rows are DEVICE arrays [B] (alphas_cumprod[t]**0.5 and (1 - alphas_cumprod[t])**0.5 as
 * fp32, built by the caller), per_sample = C*H*W.  Two products and one sum in fp32, no FMA: bit-equal to torch.        */
int sisic_add_noise(sisic_ctx*, const float* x0, const float* noise, const float* sqrt_alpha_prod,
                    const float* sqrt_one_minus_alpha_prod, float* out, int B, int64_t per_sample, void* stream);
/* noise_pred = model(noisy, timesteps).sample in training mode (train_diffusion.py:218): the inference kernels, every
 * activation and GroupNorm statistic kept for the backward pass.  timesteps: host int64 [B], one per sample.
 * `sample` is read again by sisic_unet_backward (conv_in's weight gradient): it must stay valid until then.          */
int sisic_unet_train_forward(sisic_unet*, const float* sample, const int64_t* timesteps, float* out, int B, int H, int W,
                             void* stream);
/* F.mse_loss(pred, target) (train_diffusion.py:219): loss_dev[0] = mean((pred - target)^2) (NULL: kept internally),
 * dpred = grad_scale * 2 (pred - target) / n (NULL: loss only).  Fixed-order reduction.                               */
int sisic_mse_loss(sisic_unet*, const float* pred, const float* target, int64_t n, float grad_scale, float* loss_dev,
                   float* dpred, void* stream);
/* loss.backward() (train_diffusion.py:231): dout = d loss / d model output [B,C,H,W]; fills the gradient arena
 * (overwrites: one backward per step, like zero_grad(set_to_none=True) + backward) and releases the tape.              */
int sisic_unet_backward(sisic_unet*, const float* dout, void* stream);
int sisic_unet_zero_grad(sisic_unet*, void* stream);
/* scaler.step(optimizer) (train_diffusion.py:232-233) with torch.optim.Adam's update: gradients are multiplied by inv_scale,
 * m/v/step advance, parameters move, and every packed form of the weights is rebuilt.  lr, betas and eps are doubles, as
 * torch.optim.Adam holds them (1 - beta and lr / bias_correction are formed in double and rounded to fp32 once).  found_inf != NULL: the gradients are
 * first checked for inf/nan (one synchronisation); *found_inf = 1 skips the update like GradScaler.step does.           */
int sisic_unet_optimizer_step(sisic_unet*, double lr, double beta1, double beta2, double eps, float inv_scale, int* found_inf,
                              void* stream);
/* The whole loop body (train_diffusion.py:215-233) on one stream: add_noise, forward, MSE, backward, optimizer step.
 * images / noise: dev [B,C,H,W]; timesteps and the two add_noise coefficient rows: HOST arrays [B].
 * loss_out (host, may be NULL) receives the unscaled loss (one synchronisation).                                       */
int sisic_unet_train_step(sisic_unet*, const float* images, const float* noise, const int64_t* timesteps,
                          const float* sqrt_alpha_prod, const float* sqrt_one_minus_alpha_prod, int B, int H, int W, double lr,
                          double beta1, double beta2, double eps, float loss_scale, float* loss_out, int* found_inf, void* stream);
/* Copy one tensor of the state dict (index as in sisic_unet_tensor_name) to the host: what = 0 parameter, 1 gradient,
 * 2 Adam first moment, 3 Adam second moment.  Synchronises the device.                                                  */
int sisic_unet_read(sisic_unet*, int what, int index, float* host_out, int64_t numel);
int64_t sisic_unet_train_steps(const sisic_unet*);

/* Single-operator entry points of the backward pass (parity-test surface).
 * dW of a convolution: arguments as sisic_conv_args (prologue and index maps of the FORWARD convolution), dy = gradient of
 * its output, dw = OIHW [Cout, c0+c1, k, k].                                                                            */
int sisic_conv2d_wgrad(sisic_ctx*, const sisic_conv_args* fwd_args, const float* dy, float* dw, void* stream);
/* attention backward: dqkv [B,3C,N] from qkv, the forward output o [B,C,N] and its gradient dO.                         */
int sisic_attention_bwd(sisic_ctx*, const float* qkv, const float* o, const float* dO, float* dqkv, int B, int C, int N,
                        int head_dim, void* stream);
/* GroupNorm(+SiLU) backward: a = act(GroupNorm(x)); given da, ADDS dx to dx_accum and writes dgamma, dbeta.
 * scale/shift: outputs of sisic_groupnorm_stats for x.                                                                  */
int sisic_groupnorm_bwd(sisic_ctx*, const float* da, const float* x, int B, int C, int HW, int groups, float eps,
                        const float* gamma, const float* beta, int silu, float* dx_accum, float* dgamma, float* dbeta,
                        void* stream);

/* ---- ResNet18 classifier (xai/XAI.py:357-471, forward only) ---------------------------- */
/* torchvision resnet18 with fc -> num_classes, eval mode (BatchNorm folded at load).  The state dict
 * uses the reference's key names, prefixed "model." (XAI.py:389), float tensors only: conv/fc weights
 * and biases, BatchNorm weight/bias/running_mean/running_var (num_batches_tracked is not a float tensor
 * and is not passed).                                                                         */
int sisic_resnet_create(sisic_ctx*, int num_classes, sisic_resnet** out);
int sisic_resnet_destroy(sisic_resnet*);
int sisic_resnet_num_tensors(const sisic_resnet*);
const char* sisic_resnet_tensor_name(const sisic_resnet*, int index);
int sisic_resnet_load(sisic_resnet*, int n, const char* const* names, const float* const* host_ptrs,
                      const int64_t* numels);
/* logits[B,num_classes] = classifier.forward(x).  preprocess=1: x is dev [B,3,H,W] in [-1,1] (the
 * sampler's latents) and goes through preprocess_for_classifier (XAI.py:399-431): clamp((x+1)/2,0,1),
 * bilinear resize to 224x224 (H,W <= 224), ImageNet normalisation.  preprocess=0: x is already the
 * normalised network input [B,3,H,W].                                                         */
int sisic_resnet_forward(sisic_resnet*, const float* x, float* logits, int B, int H, int W, int preprocess,
                         void* stream);
/* relu(bn1(conv1(preprocess(x)))): the stem activation [B,64,(S+1)/2,(S+1)/2] (S = 224 with preprocess=1) the max-pool
 * routes are chosen from -- introspection for parity tests of the backward pass.                              */
int sisic_resnet_stem(sisic_resnet*, const float* x, float* c1_out, int B, int H, int W, int preprocess, void* stream);
/* d score / d x of the classifier for score = log(softmax(logits)[target] + 1e-8) (xai/XAI.py:443-459), x = the raw
 * input in [-1,1] with the pre-processing differentiated through (clamp, bilinear 224x224, normalise): the gradient
 * captum's IntegratedGradients(forward_func = get_per_class_score) and the plain-gradient fallback of
 * xai/XAI.py:1039-1109 take.  grad_x: dev [B,3,H,W]; logits_out: dev [B,n_classes] or NULL.              */
int sisic_resnet_input_gradient(sisic_resnet*, const float* x, int B, int H, int W, int target,
                                float* grad_x, float* logits_out, void* stream);

/* Grad-CAM of the class logit on model.layer4[-1].conv2 as pytorch_grad_cam's GradCAM(target_layers=[...conv2]) with
 * ClassifierOutputTarget(target) computes it in xai/XAI.py:2945-3035 (pre-processing included): cam = relu(sum_k
 * mean(dlogit/dA_k) A_k), min-max scaled, bilinearly resized to 224x224, min-max scaled again.
 * cam: dev [B,224,224]; logits_out: dev [B,n_classes] or NULL.                                            */
int sisic_resnet_gradcam(sisic_resnet*, const float* x, int B, int H, int W, int target,
                         float* cam, float* logits_out, void* stream);

/* Bytes of activation workspace the handle currently keeps resident (its pool).  The pool is sized for the last
 * (B,H,W) seen and released when the shape changes, so this does not grow with the history of batch sizes
 * (the GUI's memory poll, main.py:230-253, would otherwise watch it climb).                                  */
int64_t sisic_resnet_workspace_bytes(const sisic_resnet*);
int64_t sisic_unet_workspace_bytes(const sisic_unet*);

/* get_confidence / get_per_class_score (XAI.py:443-471): prob[b] = softmax(logits[b])[target],
 * logscore[b] = log(prob[b] + 1e-8); either output may be NULL.                               */
int sisic_class_scores(sisic_ctx*, const float* logits, int B, int n_classes, int target, float* prob,
                       float* logscore, void* stream);
/* The masked copies of compute_shap_approximation (XAI.py:1147-1161): out[s,c,y,x] = image[c,y,x] where
 * masks[s, y/patch, x/patch] != 0, else 0.  image: dev [C,H,W]; masks: dev uint8 [S,H/patch,W/patch].  */
int sisic_mask_patches(sisic_ctx*, const float* image, const uint8_t* masks, float* out, int S, int C, int H, int W,
                       int patch, void* stream);

/* ---- instrumentation (bench.py roofline leg) -------------------------------------- */
/* When enabled, every conv launch is bracketed by HIP events on its own stream and
 * accumulated per class; reading synchronises the stream.                           */
int sisic_profile_enable(sisic_ctx*, int on);
/* kind: 0 = conv3x3, 1 = conv1x1, 2 = groupnorm stats, 3 = attention, 4 = ddpm step,
 *       5 = other, 6 = the f32-MFMA Winograd kernels alone (tile_cfg 66 / 68-73 / 78 / 79), 7 = the bf16x3 Winograd kernel
 *       alone (tile_cfg 74, the dominant kernel; 6 and 7 are subsets of kind 0).  Returns accumulated milliseconds, launches,
 *       algorithmic bytes, algorithmic flops (2*MAC of the direct form) and the fp32 multiply-adds x 2 of the algorithm that
 *       ran (16/36 of the direct form for Winograd launches; kind 7 issues six bf16 products for each of them).          */
int sisic_profile_read(sisic_ctx*, int kind, double* ms, int64_t* launches, double* bytes, double* flops,
                       double* flops_executed);
int sisic_profile_reset(sisic_ctx*);

#ifdef __cplusplus
}
#endif
#endif /* SISIC_H */
