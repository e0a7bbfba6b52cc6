/* libhific_hip.so — C-ABI of the MI355X (gfx950) HiFIC hot path.
 *
 * Drop-in boundary.  The reference (Justin-Tan/high-fidelity-generative-compression) has no FFI: its hot path is
 * a chain of torch.nn modules whose arithmetic runs in ATen/cuDNN.  This library replaces exactly that arithmetic.
 * Every entry point below names the reference call site (file:line under /root/reference) whose tensor op it
 * replaces; the Python binding (hific_amd/lib.py, ctypes) is the "reference-side stub" (see INTEGRATION.md).
 *
 * Contract (all functions):
 *   - plain pointers and sizes only; every buffer is DEVICE memory owned by the caller (workspace included);
 *     the library never allocates, never synchronises, never throws; it enqueues kernels on `stream` and returns
 *   - return value: 0 ok, -1 bad argument, -2 workspace too small, -3 launch failed, -4 unsupported shape
 *   - tensors are contiguous NCHW; `dtype` is the compute/storage type of activations:
 *       HIFIC_F32 (0)  float32 storage, v_mfma_f32_32x32x2_f32   (parity mode: exact f32 fma chains)
 *       HIFIC_BF16 (1) bfloat16 storage, v_mfma_f32_32x32x16_bf16 (f32 accumulate)
 *     weights, biases, norm parameters, statistics, losses and all gradients of parameters are float32
 *   - `flags` on conv entry points, meaningful for HIFIC_BF16 only: bit0 = the input activation tensor is float32,
 *     bit1 = the output activation tensor is float32 (entropy-model boundary stays float32); forward entry points only:
 *     bit2 = C is the 3C split-bf16 reduction (hific_split3 which 0/1; profiler FLOP count only), bit3 = both operands are in
 *     the pair layout (hific_split3 which 2; C = 2 * C16, native split kernel), bits 8.. = the layer's real channel count;
 *     hific_conv2d_fwd / hific_conv2d_pack_plan(kind 0) only, with bit2: bit5 = split-in-pack - `w` is the layer's REAL float32
 *     weight [K, C / 3, R, S] ((C / 3) % 64 == 0) and the weight-pack pass forms the (hi, hi, lo) image over the 3C reduction
 *     channels itself (bit for bit what hific_split3 which = 1 + the ordinary pack produce, without the derived float32 tensor)
 *   - thread-safe for distinct streams as long as the workspaces are distinct
 *   - NO collective entry point (SURVEY section 8b proposed `hific_allreduce_bucket`): the data-parallel exchange is one
 *     all-reduce per contiguous gradient-arena slice, which is exactly ncclAllReduce(ptr, count, dtype, sum, comm, stream) -
 *     a C-ABI wrapper would add a second owner of the RCCL communicator next to torch.distributed (the process-group
 *     plumbing the brief assigns to PyTorch) and nothing else.  What the library contributes to that step is the
 *     float32 <-> bfloat16 wire conversion of a bucket (hific_cast, hific_amd.parallel payload="bf16") and the 1/world
 *     scale folded into hific_adam_apply (grad_scale).  A host that does not use torch binds ncclAllReduce directly on the
 *     same pointers ("data-parallel step").
 */
#ifndef HIFIC_HIP_H
#define HIFIC_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

#define HIFIC_F32 0
#define HIFIC_BF16 1
#define HIFIC_ACT_NONE 0
#define HIFIC_ACT_RELU 1
#define HIFIC_ACT_LEAKY 2   /* LeakyReLU(0.2), src/network/discriminator.py:44 */
#define HIFIC_PAD_ZERO 0
#define HIFIC_PAD_REFLECT 1 /* nn.ReflectionPad2d / padding_mode='reflect' */

int hific_version(void);
int hific_device_info(int device, char* name64, int* cus, int* lds_per_cu);

/* ---- tickets (round 6): two-stage reductions finished inside ONE launch --------------------------------------------------
 * Channel sums (bias gradients), scalar loss sums (MSE, BCE / least-squares GAN losses, the four entropy estimates, spectral
 * norm's <dW, W>) and the LPIPS tap sums are "partials per workgroup, then a tiny second launch" reductions (~57 second
 * launches of 4-7 us per training cycle of src/model.py:346-387; second stages of a few hundred floats).  When
 * the caller registers a ticket buffer for a stream - `bytes` / 4 unsigned counters in device memory, ZERO at registration,
 * owned by the caller, valid until removed with buf = NULL - kernels launched on that stream let their last-arriving
 * workgroup run the second stage (agent-scope release / acquire around an atomic counter; the counters are zero again when
 * the launch ends).  Without a registered buffer (or with HIFIC_TICKETS=0) every entry point keeps its two-launch form: same
 * results (the scalar, channel and tap sums bit for bit; the ChannelNorm parameter sums in another, fixed, summation order).
 * One buffer per (current device, stream) - the null stream is handle 0 on every device, so the registry is keyed by both; call
 * with the buffer's device current.  Launches of one stream are ordered, launches of different streams must not share counters.
 * 16 KiB per stream is enough for every entry point. */
int hific_set_ticket_buffer(hipStream_t stream, void* buf, size_t bytes);

/* ---- convolutions (csrc/gconv.hip) ------------------------------------------------------------------------
 * Replaces nn.Conv2d (+ the nn.ReflectionPad2d in front of it, + the activation behind it):
 *   src/network/encoder.py:56-101, src/network/generator.py:28-29,98-103,139-142, src/network/hyper.py:52-54,
 *   src/network/discriminator.py:35,53-64, torchvision AlexNet features used at
 *   src/loss/perceptual_similarity/pretrained_networks.py:59-75.
 * x [N,C,H,W], w f32 [K,C,R,S], bias f32 [K] or NULL, y [N,K,OH,OW], OH=(H+pt+pb-R)/stride+1.
 * w_scale: NULL or device pointer to one float multiplied into w while packing (1/sigma of spectral norm).
 * resid: NULL or tensor shaped like y added before the activation. */
size_t hific_conv2d_ws_bytes(int N, int C, int H, int W, int K, int R, int S, int stride, int pt, int pl, int pb,
                             int pr, int dtype);
int hific_conv2d_fwd(const void* x, const float* w, const float* w_scale, const float* bias, const void* resid,
                     void* y, int N, int C, int H, int W, int K, int R, int S, int stride, int pt, int pl, int pb,
                     int pr, int pad_mode, int act, int dtype, int flags, void* ws, size_t ws_bytes,
                     void* wcache, size_t wcache_bytes, int wcache_state, hipStream_t stream);
/* adjoint w.r.t. x (autograd of the above; includes the reflection-pad adjoint). flags: bit0 dy f32, bit1 dx f32 */
int hific_conv2d_bwd_data(const void* dy, const float* w, const float* w_scale, void* dx, int N, int C, int H,
                          int W, int K, int R, int S, int stride, int pt, int pl, int pb, int pr, int pad_mode,
                          int dtype, int flags, void* ws, size_t ws_bytes, void* wcache, size_t wcache_bytes, int wcache_state, hipStream_t stream);
/* dw f32 [K,C,R,S] (= or += when accumulate). flags: bit0 x f32, bit1 dy f32 */
int hific_conv2d_bwd_weight(const void* x, const void* dy, float* dw, int N, int C, int H, int W, int K, int R,
                            int S, int stride, int pt, int pl, int pb, int pr, int pad_mode, int accumulate,
                            int dtype, int flags, void* ws, size_t ws_bytes, hipStream_t stream);

/* Replaces nn.ConvTranspose2d: src/network/generator.py:115-137 (k3 s2 p1 op1), src/network/hyper.py:83-85
 * (k5 s2 p2 op1, k3 s1 p1).  x [N,Ci,H,W], w f32 [Ci,Co,R,S], y [N,Co,OH,OW]. */
size_t hific_conv_transpose2d_ws_bytes(int N, int Ci, int H, int W, int Co, int R, int S, int stride, int pad,
                                       int outpad, int dtype);
int hific_conv_transpose2d_fwd(const void* x, const float* w, const float* bias, void* y, int N, int Ci, int H,
                               int W, int Co, int R, int S, int stride, int pad, int outpad, int act, int dtype,
                               int flags, void* ws, size_t ws_bytes, void* wcache, size_t wcache_bytes, int wcache_state, hipStream_t stream);
int hific_conv_transpose2d_bwd_data(const void* dy, const float* w, void* dx, int N, int Ci, int H, int W, int Co,
                                    int R, int S, int stride, int pad, int outpad, int dtype, int flags, void* ws,
                                    size_t ws_bytes, void* wcache, size_t wcache_bytes, int wcache_state, hipStream_t stream);
int hific_conv_transpose2d_bwd_weight(const void* x, const void* dy, float* dw, int N, int Ci, int H, int W,
                                      int Co, int R, int S, int stride, int pad, int outpad, int accumulate,
                                      int dtype, int flags, void* ws, size_t ws_bytes, hipStream_t stream);

/* ---- ChannelNorm2D (csrc/norm.hip) — src/normalisation/channel.py:48-59 (+ the ReLU after it) ------------- */
int hific_channelnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                          int N, int C, int HW, float eps, int relu, int dtype, hipStream_t stream);
/* y = act(norm(x)) + resid: the residual add of a ResidualBlock (src/network/generator.py:44) folded into the block's second
 * norm (both terms rounded to the storage type first, i.e. bit-identical to hific_channelnorm_fwd followed by hific_add).
 * Returns -4 when the shape has no register-resident configuration (callers then add separately). */
int hific_channelnorm_fwd_res(const void* x, const float* gamma, const float* beta, const void* resid, void* y, float* mean,
                              float* rstd, int N, int C, int HW, float eps, int relu, int dtype, hipStream_t stream);
/* The norm between two split-bf16 convolutions of the exact-index chain (encoder.py:56-93 blocks in bf16 mode): z is the
 * float32 output of the exact convolution; writes zb = bf16(z) (for this norm's backward), y = bf16 of the float32 result
 * (the nominal activation of the bf16 autograd graph) and x3 [N,3C,HW] = (hi, lo, hi) of that result (operand of the next
 * exact convolution, hific_split3 layout).  split_layout 0: x3 as above; 2: x3 is [N, 2 * C16, HW] in the pair layout of
 * the native split kernels (hific_split3 which = 2).  Returns -4 when the shape has no register-resident configuration or
 * a split image would exceed 2^31 elements per sample (callers then run the plain bf16 chain). */
int hific_channelnorm_fwd_exact(const float* z, const float* gamma, const float* beta, void* zb, void* y, void* x3,
                                float* mean, float* rstd, int N, int C, int HW, float eps, int relu, int split_layout,
                                hipStream_t stream);
/* The second norm of a ResidualBlock inside the exact Generator chain (src/network/generator.py:37-44 in bf16 mode with
 * ops.set_exact_training / set_exact_reconstruction): y = norm(z) + resid, the residual read as hi + lo from the split image
 * resid3 of the block's input (resid_layout 0: (hi, lo, hi) over 3C channels, 2: pair layout) so that the float32-accurate
 * trunk value never exists as a float32 tensor; outputs and return codes as hific_channelnorm_fwd_exact. */
int hific_channelnorm_fwd_exact_res(const float* z, const float* gamma, const float* beta, const void* resid3, int resid_layout,
                                    void* zb, void* y, void* x3, float* mean, float* rstd, int N, int C, int HW, float eps,
                                    int relu, int split_layout, hipStream_t stream);
size_t hific_channelnorm_bwd_ws_bytes(int N, int C, int HW);
/* dprev_bias (nullable, f32 [C]): additionally (=|+= by accumulate_prev) sum_{n,hw} dx, i.e. the bias gradient of the
 * convolution whose output is x when this norm is its only consumer (encoder.py:56-93, generator.py:28-42,115-137): saves
 * that layer's separate reduction pass over its gradient tensor. */
int hific_channelnorm_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const float* mean,
                          const float* rstd, void* dx, float* dgamma, float* dbeta, int N, int C, int HW, int relu,
                          int accumulate, int dtype, void* ws, size_t ws_bytes, float* dprev_bias, int accumulate_prev,
                          hipStream_t stream);

/* ---- elementwise / reductions (csrc/elementwise.hip) -------------------------------------------------------- */
/* dx = y>0 ? dy : slope*dy — backward of F.relu (src/network/hyper.py:59-60,91-92) / LeakyReLU */
int hific_act_bwd(const void* dy, const void* y, void* dx, long long n, float slope, int dtype, hipStream_t stream);
/* `normalize_input_image` option (src/model.py:155-156 tanh on the reconstruction; :206-209, :338-340, :361-363 the
 * [-1,1] -> [0,1] map): y = tanh(x); dx = dy (1 - y^2); y = a x + b. */
int hific_tanh_fwd(const void* x, void* y, long long n, int dtype, hipStream_t stream);
int hific_tanh_bwd(const void* y, const void* dy, void* dx, long long n, int dtype, hipStream_t stream);
int hific_scale_shift(const void* x, void* y, long long n, float a, float b, int dtype, hipStream_t stream);
/* residual adds: src/network/generator.py:44,161; also gradient fan-in sums */
int hific_add(const void* a, const void* b, void* o, long long n, int dtype, hipStream_t stream);
int hific_cast(const void* a, int src_dtype, void* o, int dst_dtype, long long n, hipStream_t stream);
/* Split-bf16 operands of the exact-index mode: the Encoder -> analysis -> synthesis_mu chain whose output is floored
 * into the latent indices (src/hyperprior.py:68-74,108-122; src/network/encoder.py:104-111) runs its forward
 * contractions as x*w ~= xh*wh + xl*wh + xh*wl (hi = bf16(v), lo = bf16(v - hi)), i.e. over 3C reduction channels of the
 * ordinary bf16 kernels.  src f32 [outer][C][inner] -> dst [outer][3C][inner]; which 0 = activation layout (hi, lo, hi),
 * 1 = weight layout (hi, hi, lo); dst_dtype HIFIC_BF16 or HIFIC_F32 (values exactly bf16-representable).
 * which 2 = the PAIR layout of the native split kernels (round 4): dst [outer][2 * C16][inner], C16 = C rounded up to 16,
 * source channel 16 g + j -> channels 32 g + j (hi) and 32 g + 16 + j (lo), padding channels zero; activations and weights
 * alike.  A convolution whose flags carry bit 3 reads both operands in this layout and issues hi*hi + hi*lo + lo*hi per
 * 16-channel slice pair itself: 2C instead of 3C staged channels for the same three MFMAs. */
/* Reflect (ReflectionPad2d) or zero padding of [planes, H, W] into [planes, H + pt + pb, W + pl + pr]: the EVALUATION path pads
 * images and latents to a multiple of 16 / 4 before compressing (src/helpers/utils.py:50-62, src/model.py:276-284). */
int hific_pad2d(const void* x, void* y, long long planes, int H, int W, int pt, int pl, int pb, int pr, int reflect, int dtype,
                hipStream_t stream);
/* Sum of two activations held as split-bf16 images (the head skip of the exact Generator chain, src/network/generator.py:161):
 * y3 (layout lo) and the nominal bf16 y [N,C,HW] of (a_hi + a_lo) + (b_hi + b_lo); layouts 0 = (hi, lo, hi) over 3C channels,
 * 2 = pair layout. */
int hific_add_split(const void* a3, int la, const void* b3, int lb, void* y, void* y3, int lo, int N, int C, int HW,
                    hipStream_t stream);
int hific_split3(const float* src, void* dst, long long outer, int C, long long inner, int which, int dst_dtype,
                 hipStream_t stream);
int hific_axpby_f32(const float* a, const float* b, float* o, float alpha, float beta, long long n,
                    hipStream_t stream);
/* out[c] = sum_{n,hw} x[n,c,hw] (bias gradients) */
int hific_channel_sum(const void* x, float* out, int N, int C, int HW, int accumulate, int dtype, void* ws,
                      size_t ws_bytes, hipStream_t stream);
/* nn.MaxPool2d(3, 2) of torchvision AlexNet (pretrained_networks.py:59) */
/* nn.MaxPool2d(2, 2) of torchvision VGG16.features (LPIPS net='vgg', pretrained_networks.py:96-134): y [planes, H/2, W/2];
 * backward routes each window's gradient to its first maximum (torch semantics). */
int hific_maxpool2s2_fwd(const void* x, void* y, long long planes, int H, int W, int dtype, hipStream_t stream);
int hific_maxpool2s2_bwd(const void* x, const void* dy, void* dx, long long planes, int H, int W, int dtype,
                         hipStream_t stream);
int hific_maxpool3s2_fwd(const void* x, void* y, long long planes, int H, int W, int dtype, hipStream_t stream);
int hific_maxpool3s2_bwd(const void* x, const void* dy, void* dx, long long planes, int H, int W, int dtype,
                         hipStream_t stream);
/* The scalar loss composition of one training forward as one launch (src/model.py:211-220,373-376; src/loss/losses.py:8-28):
 * total = ((penalty * nbpp + k_M * mse) + k_P * mean_b lp[b]) [+ beta * g_loss], penalty = q > target ? lambda_A : lambda_B,
 * all operands device scalars (lp: B floats, g_loss nullable, q = the - under data parallelism globally averaged - q_bpp);
 * aux[4] = {perceptual, penalty, penalty * nbpp, k_M * mse}.  Backward: grads[3 + B] = g * d total / d {mse, nbpp, g_loss,
 * lp[0..B)}.  Replaces ~12 zero-dimensional ATen launches and autograd nodes per forward. */
int hific_loss_combine_fwd(const float* mse, const float* lp, int B, const float* nbpp, const float* q, const float* g_loss,
                           float kM, float kP, float lamA, float lamB, float target, float beta, float* total, float* aux,
                           hipStream_t stream);
int hific_loss_combine_bwd(const float* g, const float* aux, int B, float kM, float kP, float beta, float* grads,
                           hipStream_t stream);
/* distortion loss mean((255 a - 255 b)^2): src/model.py:190-194 */
int hific_mse_fwd(const void* a, const float* b, float* out, long long n, float scale, int dtype, void* ws,
                  size_t ws_bytes, hipStream_t stream);
int hific_mse_bwd(const void* a, const float* b, const float* g, void* da, long long n, float scale, int dtype,
                  hipStream_t stream);
/* F.binary_cross_entropy_with_logits vs ones/zeros: src/loss/losses.py:30-41 */
int hific_bce_fwd(const float* z, float target, float* out, long long n, void* ws, size_t ws_bytes,
                  hipStream_t stream);
int hific_bce_bwd(const float* z, float target, const float* g, float* dz, long long n, int accumulate,
                  hipStream_t stream);
/* least-squares GAN loss on the sigmoid output, src/loss/losses.py:43-50 (`D_real`, `D_gen` = sigmoid(logits),
   src/network/discriminator.py:84-85): out[0] = mean (sigmoid(z) - target)^2 and its gradient w.r.t. the logits */
int hific_lsq_sigmoid_fwd(const float* z, float target, float* out, long long n, void* ws, size_t ws_bytes,
                          hipStream_t stream);
int hific_lsq_sigmoid_bwd(const float* z, float target, const float* g, float* dz, long long n, int accumulate,
                          hipStream_t stream);
int hific_sigmoid_f32(const float* z, float* o, long long n, hipStream_t stream);   /* discriminator.py:84 */
/* torch.cat((x, nn.Upsample(16,'nearest')(y)), 1): src/network/discriminator.py:36,75-77 */
int hific_upcat_fwd(const void* img, const void* ctx, void* out, int N, int Ci, int Cc, int H, int W, int f,
                    int dtype, hipStream_t stream);
int hific_upcat_bwd(const void* dout, void* dimg, int n0, int nimg, void* dctx, int N, int Ci, int Cc, int H, int W,
                    int f, int dtype, hipStream_t stream);
/* The Discriminator input of a G / D turn from its three sources, without materialising torch.cat([x_real, x_gen]) and
 * repeat_interleave(latents, 2) (src/model.py:176-179): image n < B is real[n], image n >= B is gen[n - B], image n reads the
 * context map ctx[n >> 1] (the reference's latent-pairing quirk), so the context conv runs once per latent.  bwd: dgen =
 * dout[B:, :Ci] and dctx[k] = block sums of dout[2k, Ci:] + dout[2k+1, Ci:]; either may be null. */
int hific_upcat_pair_fwd(const void* real, const void* gen, const void* ctx, void* out, int B, int Ci, int Cc, int H, int W,
                         int f, int dtype, hipStream_t stream);
int hific_upcat_pair_bwd(const void* dout, void* dgen, void* dctx, int B, int Ci, int Cc, int H, int W, int f, int dtype,
                         hipStream_t stream);
/* Gradient of the per-latent context maps through the Discriminator's first convolution (4x4, stride 2, reflect pad 1 on
 * cat(image, upsample_f(context)); src/network/discriminator.py:53,75-78) straight from dz = the gradient at that convolution's
 * pre-activation [2B, K, H/2, W/2]: dctx[m][c][Y][X] = (1/sigma) sum_{i in 2m,2m+1} sum_k sum_{r,s} w[k][Ci+c][r][s] * (window sum
 * of dz[i][k] over the output pixels whose tap (r, s) reads the f x f block (Y, X), mirror row / column included).  Replaces the
 * 15-channel data gradient on the padded plane + hific_upcat_pair_bwd's block sums by one pass over dz and a small contraction.
 * w = weight_orig [K, Ci+Cc, 4, 4] float32, inv_sigma: device scalar or null; ws: B * (H/f) * (W/f) * K * 16 floats.
 * HIFIC_ERR_UNSUPPORTED: anything but the reference's layer (f = 16, K = 64, Cc = 12). */
int hific_d1_ctx_grad(const void* dz, const float* w, const float* inv_sigma, void* dctx, int B, int K, int Ci, int Cc, int H,
                      int W, int f, int dtype, void* ws, size_t ws_bytes, hipStream_t stream);
/* The same power iteration for n <= 8 layers in one call (src/network/discriminator.py:53-62: the four spectral-norm convs of
 * the Discriminator; their iterations depend only on the weights): per layer bit-identical to hific_spectral_norm_fwd, 6
 * launches for the whole set instead of 6 per layer.  ws >= sum_i (M_i + K_i + ceil(K_i / 16) M_i) floats.
 * do_iter bit 0: run the iteration (training); bit 1 (round 6): sig[i] has 2 + K_i + M_i floats and receives, behind
 * [sigma, 1/sigma], copies of the post-iteration u_i and v_i - what the backward of THIS forward needs once the next forward
 * has iterated u, v in place (torch.nn.utils.spectral_norm clones them for the same reason). */
int hific_spectral_norm_fwd_batch(const float* const* W, float* const* u, float* const* v, float* const* sig, const int* K,
                                  const int* M, int n, int do_iter, float eps, void* ws, size_t ws_bytes, hipStream_t stream);
/* torch.nn.utils.spectral_norm power iteration + sigma (discriminator.py:46-62); sigma_out = {sigma, 1/sigma} */
int hific_spectral_norm_fwd(const float* W, float* u, float* v, float* sigma_out, int K, int M, int do_iter,
                            float eps, void* ws, size_t ws_bytes, hipStream_t stream);
int hific_spectral_norm_bwd(const float* dW, const float* Worig, const float* u, const float* v,
                            const float* sigma, float* dWorig, int K, int M, int accumulate, void* ws,
                            size_t ws_bytes, hipStream_t stream);
/* torch.optim.Adam step over a flat arena (train.py:287-301: lr 1e-4, betas (.9,.999), eps 1e-8, no decay) */
int hific_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                    float eps, int step, float grad_scale, hipStream_t stream);
/* The same update with the step count in device memory, so that a captured hipGraph of the training step can be replayed
 * (kernel arguments are frozen at capture): hific_adam_prepare advances *step_dev and writes the two bias-correction
 * factors to bc_dev[0..1]; hific_adam_apply updates one parameter range with them. */
int hific_adam_prepare(int* step_dev, float* bc_dev, float beta1, float beta2, hipStream_t stream);
int hific_adam_apply(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                     float eps, const float* bc_dev, float grad_scale, hipStream_t stream);

/* ---- entropy model (csrc/entropy.hip), all float32 ----------------------------------------------------------- */
/* floor(x - mean + .5) + mean: src/hyperprior.py:68-74,108-122 (mean may be NULL) */
int hific_round_f32(const float* x, const float* mean, float* o, long long n, hipStream_t stream);
/* EVALUATION path, device half of `compress` - int32 outputs for the host rANS coder:
 * symbols = floor(y + 0.5 - mean) and indices = table entry of max(scale, scales_min)
 * (src/compression/prior_model.py:148-156,180-181; table = `scale_table_tensor`, n_table <= 256) */
int hific_prior_symbols(const float* x, const float* mean, const float* scale, const float* table, int n_table,
                        float scales_min, int* symbols, int* indices, long long n, hipStream_t stream);
/* symbols = floor(z + 0.5), indices = channel number (src/compression/hyperprior_model.py:135-139,169); z [N,C,HW] */
int hific_hyper_symbols(const float* z, int* symbols, int* indices, int N, int C, int HW, hipStream_t stream);
/* LowerBoundToward: src/helpers/maths.py:87-100 */
int hific_lower_bound_fwd(const float* x, float bound, float* o, long long n, hipStream_t stream);
int hific_lower_bound_bwd(const float* x, const float* dy, float bound, float* dx, long long n, hipStream_t stream);
/* mul * sum(log(p + eps)): src/hyperprior.py:80-93 */
int hific_logsum_fwd(const float* p, float* out, long long n, float eps, float mul, void* ws, size_t ws_bytes,
                     hipStream_t stream);
int hific_logsum_bwd(const float* p, const float* g, float* dp, long long n, float eps, float mul, int accumulate,
                     hipStream_t stream);
/* latent_likelihood: src/hyperprior.py:124-139 with maths.py:102-109 CDFs (logistic=1 selects sigmoid) */
int hific_gauss_lik_fwd(const float* x, const float* mean, const float* scale, float* lik, long long n,
                        float min_lik, int logistic, hipStream_t stream);
int hific_gauss_lik_bwd(const float* x, const float* mean, const float* scale, const float* dlik, float* dx,
                        float* dmean, float* dscale, long long n, float min_lik, int logistic, int acc_mean,
                        int acc_scale, hipStream_t stream);
/* HyperpriorDensity.likelihood: src/compression/hyperprior_model.py:305-326,349-384.
 * params / dparams: 12 device pointers H_0..H_3, a_0..a_3, b_0..b_3 (reference parameter shapes) */
int hific_factorized_lik_fwd(const float* x, const float* const* params, float* lik, int N, int C, int HW,
                             float min_lik, hipStream_t stream);
int hific_factorized_lik_bwd(const float* x, const float* const* params, const float* dlik, float* dx,
                             float* const* dparams, int N, int C, int HW, float min_lik, int accumulate, void* ws,
                             size_t ws_bytes, hipStream_t stream);

/* ---- LPIPS taps (csrc/lpips.hip) — perceptual_loss.py:36-46, networks_basic.py:61-108 ------------------------- */
int hific_lpips_prep(const void* src0, int s0_f32, const void* src1, int s1_f32, void* out, int B, int HW,
                     int normalize, int dtype, hipStream_t stream);
int hific_lpips_prep_bwd(const void* dgen /* [B,3,HW]: pred half only */, void* dsrc1, int B, int HW, int normalize, int dtype, int out_f32,
                         hipStream_t stream);
int hific_lpips_tap_fwd(const void* f, const float* w, float* val, int B, int C, int HW, int accumulate, int dtype,
                        void* ws, size_t ws_bytes, hipStream_t stream);
int hific_lpips_tap_bwd(const void* f, const float* w, const float* gval, void* df1, int B, int C, int HW,
                        int accumulate, int dtype, hipStream_t stream);

/* ---- persistent packed-weight cache ---------------------------------------------------------------------------
 * The MFMA kernels read weights from a packed bf16/f32 image [K_pad][tap][C_pad] (per stride phase).  Without a cache
 * every forward-type call re-packs its f32 weights into the workspace first.  With one, the caller owns a buffer per
 * (weight tensor, direction, geometry) and passes it as `wcache`:
 *   wcache_state 0: no cache (pack into the workspace);  1: pack into wcache now;  2: wcache is current - skip the pack.
 * hific_*_pack_plan fills an opaque host-side job (hific_pack_job_bytes() bytes) describing exactly the packing that
 * the matching entry point (kind 0 = forward, 1 = data gradient) would do for that geometry; after
 * hific_pack_job_set_ptrs (destination = the cache buffer, source = the f32 weights, optional device scalar scale)
 * the jobs are copied to device memory and hific_pack_batch re-packs ALL of them in one launch (after an optimizer
 * step): prefix_dev[j] = first block of job j (prefix of hific_pack_job_info's nblocks), lds_bytes = max over jobs. */
size_t hific_pack_job_bytes(void);
int hific_conv2d_pack_plan(int kind, int N, int C, int H, int W, int K, int R, int S, int stride, int pt, int pl, int pb,
                           int pr, int pad_mode, int dtype, int flags, void* job, size_t job_bytes);
int hific_conv_transpose2d_pack_plan(int kind, int N, int Ci, int H, int W, int Co, int R, int S, int stride, int pad,
                                     int outpad, int dtype, int flags, void* job, size_t job_bytes);
int hific_pack_job_set_ptrs(void* job, void* wpack, const float* w, const float* w_scale);
int hific_pack_job_info(const void* job, int* nblocks, int* lds_bytes, long long* wpack_bytes, int* dtype);
int hific_pack_batch(const void* jobs_dev, const int* prefix_dev, int njobs, int total_blocks, size_t lds_bytes, int dtype,
                     hipStream_t stream);

/* ---- training-time augmentation (csrc/augment.hip) - src/helpers/datasets.py:206-216 ----------------------------
 * RandomHorizontalFlip -> Resize((ceil(s H), ceil(s W)), PIL bilinear) -> RandomCrop(crop) -> ToTensor [-> Normalize]
 * for a batch of decoded uint8 HWC images in one launch; integer-exact with Pillow's 8-bit resampler.  The random
 * draws and the fixed-point weights of the crop window (Resample.c precompute_coeffs, double precision) come from the
 * host (hific_amd/helpers/augment.py).  imgs: device array of B hific_aug_image; xb/yb: int32 [B][crop][2] = (first
 * source index, tap count) per crop column/row; xk/yk: int32 [B][crop][kmax] weights (22 fractional bits);
 * out: float32 [B,3,crop,crop]. */
typedef struct hific_aug_image {
    const unsigned char* src;   /* uint8 [H][W][3], device memory */
    int H, W;
    int flip;                   /* horizontal flip before the resize */
    int resize_x, resize_y;     /* 0: the axis keeps its size (no resampling pass, as in Pillow) */
    int top, left;              /* crop origin in the resized image */
} hific_aug_image;
int hific_augment_crop(const void* imgs, const int* xb, const int* xk, const int* yb, const int* yk, int B, int crop,
                       int kmax, int normalize, float* out, hipStream_t stream);

/* ---- in-library profiler (bench.py roofline) ------------------------------------------------------------------
 * Measurement facility, OFF unless hific_prof_begin() was called, and the one exception to the contract above: it
 * keeps process-global state (an event pool created on demand with hipEventCreate, never freed), is not thread-safe
 * and must not be enabled during hipGraph capture.  Between begin and end every GEMM-class launch is bracketed by
 * an event pair on the launch stream and keyed by its KERNEL FUNCTION (e.g. "gconv_sp9_kernel<2>", "wgrad_pipe_kernel").
 * hific_prof_end synchronises and returns, per kernel function (at most max_kinds): total ms, total ALGORITHMIC
 * FLOPs (2*MACs of the op on its real output domain), launch count, name (max_kinds x 64 chars, NUL-terminated).
 * Return value: number of kernel functions seen (>= 0) or a negative error code. */
int hific_prof_begin(void);
int hific_prof_end(int max_kinds, double* ms, double* flops, int* count, char* names);
/* Algorithmic HBM bytes (operands read once + result written once) per kernel function of the profile in progress, same order
 * as hific_prof_end; call it BEFORE hific_prof_end (which resets the profile).  0 for launches that do not report bytes. */
int hific_prof_bytes(int max_kinds, double* bytes);

/* The planner's HIFIC_* environment knobs (A/B switches of the conv engine; the reference has no counterpart - its knobs are
 * cuDNN's) are read once per process and cached.  A caller that changes one inside a running process (tests, tools) calls this
 * to drop the cache; packed weights made under the old setting must be dropped by the caller too. */
int hific_env_refresh(void);

#ifdef __cplusplus
}
#endif
#endif
