/* dct.h -- C ABI of the MI355X-native co-training step library (libdct_hip.so).
 *
 * Drop-in boundary: the reference has no FFI; its hot path is Python calling
 * third-party ATen ops.  This header declares one entry point per ATen op family the step
 * invokes; each comment cites the reference call site it replaces.
 * The Python host (package `dct_amd`) binds these with ctypes and puts them under the
 * reference's own object API (CoTrainer / Segmentator / loss modules / FSGMGenerator).
 *
 * Conventions
 *   - device pointers only; the library never allocates or frees device memory; workspaces are
 *     caller-provided (size from the matching *_workspace_bytes);
 *   - tensors are NHWC *views*: channel stride 1, element strides for n/h/w (so channel slices
 *     of a concat buffer and interior windows of padded buffers are plain views);
 *   - every launch goes to the hipStream_t passed in (void* here: no HIP headers needed);
 *   - return 0 on success, negative dct_status on failure; no exceptions cross the ABI;
 *   - the compute entry points keep no state between calls and are re-entrant from any thread.  The ONLY process-global
 *     mutable state is diagnostic: the A/B tuning knobs (dct_tune_set, defaults = the shipped configuration; read at
 *     launch time, so set them between launches from one thread) and the per-class event timing (dct_prof_enable /
 *     dct_prof_read).  SURVEY.md 8b sketched a `dct_init(device)` handle; it was dropped on purpose: every launch
 *     already names its device through the stream it is given and the pointers it receives, and nothing else (no
 *     allocation, no cached workspace, no per-device table) would live in such a handle.
 */
#ifndef DCT_H_
#define DCT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum dct_status {
  DCT_OK = 0,
  DCT_ERR_BAD_ARG = -1,      /* null pointer, negative size, inconsistent shapes */
  DCT_ERR_UNSUPPORTED = -2,  /* shape/dtype outside what the kernels are built for */
  DCT_ERR_LAUNCH = -3,       /* hipGetLastError() != hipSuccess after a launch */
  DCT_ERR_WORKSPACE = -4     /* workspace too small */
} dct_status;

/* DCT_F16 (IEEE half, BASELINE configs[4] "fp16"): storage type of the Enet kernels' activations and activation gradients
 * (dct_enet_*), with fp32 raw conv outputs / statistics / parameters exactly as in bf16 mode.  Half's 5-bit exponent cannot
 * hold per-pixel loss gradients of a mean over ~1e6 pixels, so the host scales the loss gradients by a power of two (the
 * `gmul` argument of the loss backward kernels) and the optimizer divides it out (`grad_scale` of dct_adam_flat*): exact,
 * because a power-of-two factor commutes with every rounding on the way.  The MFMA UNet kernels take F32 / BF16 only. */
typedef enum dct_dtype { DCT_F32 = 0, DCT_BF16 = 1, DCT_F16 = 2 } dct_dtype;

/* NHWC strided view; strides in ELEMENTS of the view's dtype; channel stride is 1. */
typedef struct dct_view {
  void* ptr;
  int32_t n, h, w, c;
  int64_t sn, sh, sw;
} dct_view;

typedef void* dct_stream; /* hipStream_t */

int dct_version(void);
const char* dct_status_string(int status);

/* ---- K1/K2/K3: convolutions as implicit GEMM on MFMA -----------------------------------
 * Replaces F.conv2d / F.conv_transpose2d behind nn.Conv2d / nn.ConvTranspose2d
 * (arch/network.py:120-223, arch/enet.py:21-122) and their autograd backward.
 *
 * y[n,oy,ox,co] = epi( sum_{r,s,ci} x[n, oy*stride + r*dil - pad_h, ox*stride + s*dil - pad_w, ci]
 *                                   * w[co][r][s][ci] )
 * w is K-major packed: [Cout][R][S][Cin] in `dtype`.  Out-of-range taps read zero.
 * epi: +bias[co] (fp32, nullable) -> relu (flag) -> * (mask[n,oy,ox,co] > 0 ? mask_scale : 0) for
 *      co < mask_channels (mask nullable; ReLU / dropout backward fused) -> (+= old y if accumulate).
 * scatter2x2 != 0: the N dimension is (a,b,co) = 4*Cout and element (pixel, (a,b,co)) is stored at
 *      y[n, 2*oy+a, 2*ox+b, co]  (ConvTranspose2d k=2 s=2 forward, network.py:126,169).
 * With dgrad-packed weights ([Cin][flipped taps][Cout]) and pad = k-1 the same entry computes
 * the input gradient of a stride-1 conv; with R=S=2, stride 2 it computes ConvTranspose2d's.
 * Cin must be a multiple of 32 (bf16) / 16 (f32); Cout (or 4*Cout) a multiple of 64.
 */
typedef struct dct_conv_desc {
  int32_t R, S, stride, dil, pad_h, pad_w;
  int32_t relu;
  int32_t scatter2x2;
  int32_t accumulate;
  int32_t mask_channels;
  float mask_scale;
  /* ReLU-gate bits: one bit per element of a DENSE bf16 NHWC tensor, byte (pixel * C/8 + c/8), bit c%8 = "element > 0".
   * relu_bits_out (nullable): dct_conv2d / dct_conv_cin1_fwd also leave the bits of y there (y dense, C % 8 == 0, bf16).
   * mask_bits (nullable, with `mask` still passed): the same information as `mask` (dense, mask_channels = all of y's
   * channels) at 1/16 of the bytes; the kernels whose epilogue can, read it instead of the activation. */
  const uint8_t* mask_bits;
  uint8_t* relu_bits_out;
  /* 2x2 / stride 2 / ceil-mode max pooling of y in the same call (nn.MaxPool2d behind the conv + ReLU of a UNet encoder block,
   * arch/network.py:120-130): pool_out (nullable) is a DENSE NHWC tensor [n][(h+1)/2][(w+1)/2][c] of y's type; pool_codes
   * (nullable) the routing codes of dct_maxpool2x2_fwd_codes, one byte per pooled element.  The shared-halo kernel pools its
   * staged output tile before the tile leaves LDS (the pool no longer re-reads y from memory; y itself is still written unless
   * pool_only says nobody reads it); every other path launches the pooling kernel behind the conv.  Results are those of
   * dct_conv2d followed by dct_maxpool2x2_fwd[_codes], bit for bit.  Not with scatter2x2 / accumulate, and not with mask /
   * mask_bits (a forward-pass feature: the staged tile is pooled before the gates of a data gradient would be applied). */
  void* pool_out;
  uint8_t* pool_codes;
  /* pool_only != 0 (with pool_out): the caller consumes only the pooled tensor -- y must still be a valid buffer, but its contents
   * after the call are unspecified: the shared-halo kernel then skips its row stores (a UNet encoder block's full-resolution output
   * is read by the pooling alone once the backward pass routes by pool_codes: 130 MB less to write at the first level). */
  int32_t pool_only;
  /* The UNet stem's weight gradient from the epilogue of the data gradient that produces its dy (network.py:159-161: Conv2d(1, 64, 3),
   * ReLU, Conv2d(64, 64, 3)): with stem_x (nullable; dense fp32 [n][y->h + 2][y->w + 2], the single-channel input of the stem) a
   * bf16 64 -> 64 channel 3x3 data gradient with mask_bits (mask_scale 1) also forms stem_dw[64][9] (+)= sum_px y[px][c] * x[px + tap] and
   * stem_db[64] (+)= sum_px y[px][c] from its masked, bf16-rounded output tile while the tile is in LDS -- and then does NOT store y
   * (132 MB written and read back per network and step, by nobody else when d/dx of the network is not asked for).  y must still be a valid
   * view (shapes).  Needs dct_conv2d_workspace_bytes of workspace; DCT_ERR_UNSUPPORTED when the layer does not take the shared-halo
   * 64-channel tile (the caller then runs dct_conv2d and dct_conv_cin1_wgrad as before). */
  const float* stem_x;
  float* stem_dw;
  float* stem_db;
  int32_t stem_accumulate;
  /* Un-pooling on load (round 5; the backward pass of a UNet encoder block, network.py:120-130: ... Conv2d, ReLU, MaxPool2d).  The gradient at the
   * block's full-resolution output is maxpool-backward of the gradient at the pooled tensor: three of four positions are zero and the
   * fourth is a copy, so it need not exist in memory.  With unpool_codes (nullable; the DENSE routing codes of dct_maxpool2x2_fwd_codes,
   * [n][(unpool_h + 1) / 2][(unpool_w + 1) / 2][c]) the input view (dct_conv2d: x; dct_conv2d_wgrad: p) is the gradient AT THE POOLED
   * tensor and the kernel expands it to the [n][unpool_h][unpool_w][c] tensor dct_maxpool2x2_bwd_codes(relu_mask = 1, scale = 1) would
   * have written while it stages -- bit for bit the result of that launch followed by the plain call.  bf16, 3x3 stride 1; dct_conv2d:
   * the data-gradient form (padding 2) on the shared-halo kernel; DCT_ERR_UNSUPPORTED (nothing launched) where the layer takes another
   * kernel: the caller then un-pools into a buffer and calls again without the codes. */
  const uint8_t* unpool_codes;
  int32_t unpool_h, unpool_w;
} dct_conv_desc;

size_t dct_conv2d_workspace_bytes(const dct_view* x, const dct_view* y, const dct_conv_desc* d, int dtype);
int dct_conv2d(const dct_view* x, const void* w_packed, const float* bias, const dct_view* mask,
               const dct_view* y, const dct_conv_desc* d, int dtype,
               void* workspace, size_t workspace_bytes, dct_stream stream);

/* Weight gradient: dw[p][r][s][q] (+)= sum_m P[m][p] * Q[pixel(m)*stride + (r,s)*dil - pad][q]
 * (fp32 output, K-major like the packed weights).  Conv2d: P = dy, Q = x.  ConvTranspose2d k2 s2:
 * P = x, Q = dy, R=S=2, stride 2.   Replaces the weight half of conv backward (autograd of
 * cotraining_totalloss.py:247). */
size_t dct_conv2d_wgrad_workspace_bytes(const dct_view* p, const dct_view* q, const dct_conv_desc* d, int dtype);
int dct_conv2d_wgrad(const dct_view* p, const dct_view* q, float* dw, const dct_conv_desc* d, int dtype,
                     void* workspace, size_t workspace_bytes, dct_stream stream);
/* Same, and db[c] (+)= sum over pixels of p[.., c] in the same launch (the bias gradient of a Conv2d is the column
 * sum of the dy tiles the kernel already holds: one extra MFMA against a ones vector).  bf16 only; db nullable. */
int dct_conv2d_wgrad_bias(const dct_view* p, const dct_view* q, float* dw, float* db, const dct_conv_desc* d, int dtype,
                          void* workspace, size_t workspace_bytes, dct_stream stream);

/* db[c] (+)= sum over pixels of dy[.., c]  (bias half of conv backward). */
int dct_bias_grad(const dct_view* dy, float* db, int accumulate, int dtype,
                  void* workspace, size_t workspace_bytes, dct_stream stream);
size_t dct_bias_grad_workspace_bytes(const dct_view* dy);
/* n (<= 8) bias gradients in one launch pair: dbs[k][c] (+)= sum over pixels of dys[k][.., c]; per job the sums and their order are those of
 * dct_bias_grad.  (A UNet's four up-convolutions: nn.ConvTranspose2d's bias gradient is the column sum of the OTHER operand of its weight-gradient
 * GEMM, so it cannot ride along there; four launch pairs of ~5 us each sat on every model's backward chain.) */
int dct_bias_grad_batched(const dct_view* dys, float* const* dbs, int n, int accumulate, int dtype,
                          void* workspace, size_t workspace_bytes, dct_stream stream);
size_t dct_bias_grad_batched_workspace_bytes(const dct_view* dys, int n);

/* Repack fp32 master weights src[P][T][Q] (T taps) into `dtype`:
 *   transpose == 0: straight cast copy;
 *   transpose == 1: dst[Q][T'][P]  (conv dgrad pack; tap order reversed when flip_taps);
 *   transpose == 2: dst[T'][Q][P]  (ConvTranspose2d forward pack for the scatter2x2 mode). */
int dct_pack_weight(const float* src, void* dst, int P, int T, int Q, int transpose, int flip_taps,
                    int dtype, dct_stream stream);
/* All transposed packs of a network in one launch.  jobs_dev: device array of njobs records
 *   { const void* src; void* dst; int32 P, T, Q, flip; int64 dq, dt; int32 tile_begin, src_bf16; }  (56 bytes;
 *     src is fp32, or the bf16 image of the same tensor when src_bf16 != 0)
 * job j covers the 32 x 32 (p, q) tiles [tile_begin[j], tile_begin[j+1]) (P/32 * Q/32 * T of them; P, Q
 * multiples of 32) of dst[q*dq + t'*dt + p] = src[p][t][q]: (dq, dt) = (T*P, P) is dct_pack_weight's
 * transpose == 1, (P, Q*P) its transpose == 2.  total_tiles = tile_begin past the last job. */
int dct_pack_weights_batched(const void* jobs_dev, int njobs, int total_tiles, int dtype, dct_stream stream);
/* The same table for 16-bit (bf16 / f16) sources AND destinations, on 64 x 64 tiles with 16-byte accesses: every job has
 * src_bf16 != 0, P and Q multiples of 64, dq and dt multiples of 8, 16-byte aligned src / dst; tile_begin counts 64 x 64
 * tiles (P/64 * Q/64 * T per job).  This is the per-step re-pack of a bf16 UNet (arch/unet.py::_ensure_packs). */
int dct_pack_weights_batched64(const void* jobs_dev, int njobs, int total_tiles, dct_stream stream);

/* First layer, Cin = 1 (network.py:159 dec1 conv; enet.py:21 initial conv): direct conv.
 * x is fp32 [N,H,W,1]; w fp32 [Cout][R][S]; y in `dtype`. */
int dct_conv_cin1_fwd(const dct_view* x, const float* w, const float* bias, const dct_view* y,
                      const dct_conv_desc* d, int dtype, dct_stream stream);
/* dx (fp32) = full correlation of dy with w  (the d/dx FGSM needs: AEGenerator.py:27-28). */
int dct_conv_cin1_dgrad(const dct_view* dy, const float* w, const dct_view* dx,
                        const dct_conv_desc* d, int dtype, dct_stream stream);
size_t dct_conv_cin1_wgrad_workspace_bytes(const dct_view* dy, const dct_conv_desc* d);
int dct_conv_cin1_wgrad(const dct_view* x, const dct_view* dy, float* dw, float* db,
                        const dct_conv_desc* d, int accumulate, int dtype,
                        void* workspace, size_t workspace_bytes, dct_stream stream);

/* Classifier head: 1x1 conv to a few classes (network.py:223 `final`), Cout <= 8.
 * x in `dtype` [.., Cin]; w fp32 [Cout][Cin]; y fp32 [.., Cout]. */
int dct_conv1x1_head_fwd(const dct_view* x, const float* w, const float* bias, const dct_view* y,
                         int dtype, dct_stream stream);
size_t dct_conv1x1_head_bwd_workspace_bytes(const dct_view* x, int cout);
/* dx (dtype, optional) = dy*W masked by (x>0) when relu_mask; dw,db (fp32, optional) (+)=. */
int dct_conv1x1_head_bwd(const dct_view* x, const dct_view* dy, const float* w, const dct_view* dx,
                         float* dw, float* db, int relu_mask, int accumulate, int dtype,
                         void* workspace, size_t workspace_bytes, dct_stream stream);

/* ---- K4: max pooling 2x2 stride 2 (ceil mode) ------------------------------------------
 * Replaces nn.MaxPool2d(2, stride=2, ceil_mode=True) (network.py:166).  y.h = ceil(x.h/2).
 * bwd: dx = dy routed to the first max in scan order, then * (x > 0 ? scale : 0) when relu_mask
 * (fuses the ReLU / dropout backward that precedes the pool). */
int dct_maxpool2x2_fwd(const dct_view* x, const dct_view* y, int dtype, dct_stream stream);
int dct_maxpool2x2_bwd(const dct_view* x, const dct_view* dy, const dct_view* dx, int relu_mask,
                       float scale, int dtype, dct_stream stream);

/* The same pair with the routing kept from the forward pass instead of re-derived from x: codes (uint8, dense
 * [N][y.h][y.w][C], 8-byte aligned) receives per pooled element bits 0-1 = window position (2*dy+dx) of the first maximum,
 * bit 2 = "maximum > 0" (the ReLU / dropout gate), bit 3 = no maximum; the backward pass reads dy and codes only. */
int dct_maxpool2x2_fwd_codes(const dct_view* x, const dct_view* y, uint8_t* codes, int dtype, dct_stream stream);
int dct_maxpool2x2_bwd_codes(const uint8_t* codes, const dct_view* dy, const dct_view* dx, int relu_mask,
                             float scale, int dtype, dct_stream stream);
/* The same with the gradient of a second reader of the pooled tensor gathered on the way: `skip` is the gradient at a bilinearly resized
 * (align_corners) copy of it -- the UNet's skip connection, network.py:205-239 `F.upsample(dec_k, size, mode='bilinear')` -- and
 * dx = unpool(dy + bilinear_backward(skip)): what dct_bilinear_bwd into dy followed by dct_maxpool2x2_bwd_codes gives, with one write and
 * one read of dy less and the sum kept in fp32 until it is routed. */
int dct_maxpool2x2_bwd_codes_skip(const uint8_t* codes, const dct_view* dy, const dct_view* skip, const dct_view* dx, int relu_mask,
                                  float scale, int dtype, dct_stream stream);

/* ---- K5: bilinear resize, align_corners=True -------------------------------------------
 * Replaces F.upsample_bilinear (network.py:232-240).  Any in/out size.  y may be a channel
 * slice of a concat buffer.  in/out dtypes given separately (the final resize reads/writes f32).
 * bwd is a gather (deterministic): dx[src] (+)= sum of dy[dst]*weight. */
int dct_bilinear_fwd(const dct_view* x, const dct_view* y, int dtype_in, int dtype_out, dct_stream stream);
/* n (<= 8) resizes of one dtype in ONE launch: ys[k] = dct_bilinear_fwd(xs[k]), bit for bit.  DCT_ERR_UNSUPPORTED for views off the 16-byte
 * vector width (use one dct_bilinear_fwd per tensor).  (A UNet's four skip connections, network.py:216-223: every pooled tensor exists once the
 * encoder is through.) */
int dct_bilinear_fwd_batched(const dct_view* xs, const dct_view* ys, int n, int dtype, dct_stream stream);
int dct_bilinear_bwd(const dct_view* dy, const dct_view* dx, int dtype_dy, int dtype_dx, int accumulate,
                     dct_stream stream);

/* ---- K9: dropout ------------------------------------------------------------------------
 * Replaces nn.Dropout(.5) (network.py:165,210).  Philox4x32-10 keyed by (seed, offset + element
 * index); y = keep ? x/(1-p) : 0.  mask_out (uint8, nullable, dense NHWC) receives the keep mask. */
int dct_dropout_fwd(const dct_view* x, const dct_view* y, uint8_t* mask_out, float p,
                    uint64_t seed, uint64_t offset, int dtype, dct_stream stream);
/* Same, with the call counter in device memory: calls = TWO 64-bit words that successive launches use in turn.  The launch reads
 * calls[parity] = n, is call number n + 1 (offset = that << 40) and stores n + 1 to calls[parity ^ 1]; the caller alternates `parity` (0, 1, 0, ...)
 * from launch to launch, stream-ordered.  Nothing in the launch depends on a per-step host value, so a captured HIP graph of the training step
 * draws a fresh mask on every replay -- provided a replay holds an EVEN number of launches per counter (a UNet forward pass has two sites). */
int dct_dropout_fwd_dev(const dct_view* x, const dct_view* y, uint8_t* mask_out, float p,
                        uint64_t seed, uint64_t* calls, int parity, int dtype, dct_stream stream);
/* dct_dropout_fwd_dev followed by dct_maxpool2x2_fwd_codes in ONE pass: y = maxpool(dropout(x)) and its routing codes, bit for bit, without the
 * dropped full-resolution tensor (the backward pass routes by the codes and the 1 / (1 - p) scale).  The fourth encoder level of a UNet's training
 * pass (network.py:160-166: conv, ReLU, Dropout, MaxPool2d).  calls / parity as dct_dropout_fwd_dev. */
int dct_dropout_maxpool2x2_fwd_codes(const dct_view* x, const dct_view* y, uint8_t* codes, float p, uint64_t seed, uint64_t* calls,
                                     int parity, int dtype, dct_stream stream);
/* y = x * (mask_u8 ? 1/(1-p) : 0) with a caller-supplied dense mask (parity replay). */
int dct_dropout_apply(const dct_view* x, const dct_view* y, const uint8_t* mask, float p, int dtype,
                      dct_stream stream);

/* ---- K7: elementwise -------------------------------------------------------------------- */
/* y = g * (a > 0 ? scale : 0)   (ReLU backward where it cannot be fused). */
int dct_relu_bwd(const dct_view* g, const dct_view* a, const dct_view* y, float scale, int dtype,
                 dct_stream stream);
int dct_cast(const dct_view* x, const dct_view* y, int dtype_in, int dtype_out, dct_stream stream);

/* ---- K10: pixel-wise losses on fp32 NHWC logits [P pixels][C], 2 <= C <= 8 ----------------
 * CE     : loss/loss.py:12-25   (log_softmax + NLL, mean over targets != ignore_index)
 * softmax: models/segmentators.py:50
 * JSD    : loss/loss.py:183-196 + :70-84 ; fused from logits of S <= 4 models; *mean* over pixels
 * KL     : loss/loss.py:110-134 ; KL(y || p), y detached, mean over pixels
 * All reductions are two-stage and deterministic.  Scalars are written to device memory;
 * nothing synchronises with the host.
 */
size_t dct_loss_workspace_bytes(int64_t pixels);
/* out[0] = mean CE, out[1] = number of counted pixels.  targets int64. */
int dct_ce_fwd(const float* logits, const int64_t* targets, int64_t pixels, int C, int ignore_index,
               float* out2, void* workspace, size_t workspace_bytes, dct_stream stream);
/* dlogits (=|+=) gscale[0] * gmul * (softmax - onehot) / count ; gscale is a device scalar (nullable=1) */
int dct_ce_bwd(const float* logits, const int64_t* targets, int64_t pixels, int C, int ignore_index,
               const float* count, const float* gscale, float gmul, float* dlogits, int accumulate,
               dct_stream stream);
/* dct_ce_fwd + dct_ce_bwd of the same logits in two launches instead of three: the backward kernel folds the forward kernel's block partials
 * itself (every block, in the forward call's order: out2 and dlogits are bit for bit those of the two calls).  The co-training step's supervised
 * term (cotraining_totalloss.py:211-216 + the loss.backward() of :247). */
int dct_ce_step(const float* logits, const int64_t* targets, int64_t pixels, int C, int ignore_index, float* out2,
                const float* gscale, float gmul, float* dlogits, int accumulate, void* workspace, size_t workspace_bytes,
                dct_stream stream);
int dct_softmax_fwd(const float* logits, float* probs, int64_t pixels, int C, dct_stream stream);
/* dlogits (=|+=) p * (dprobs - sum_c dprobs*p) */
int dct_softmax_bwd(const float* probs, const float* dprobs, float* dlogits, int64_t pixels, int C,
                    int accumulate, dct_stream stream);
/* map[pix] = -sum_c p*log(p+1e-16) */
int dct_entropy_fwd(const float* probs, float* map, int64_t pixels, int C, dct_stream stream);
/* dprobs[pix][c] = dmap[pix] * -(log(p+1e-16) + p/(p+1e-16)) */
int dct_entropy_bwd(const float* probs, const float* dmap, float* dprobs, int64_t pixels, int C, dct_stream stream);
/* probs-in variants keep the reference module API (JSD_2D()(list_of_probs) -> [B,H,W] map). */
int dct_jsd_map_fwd(const float* const* probs, int S, float* map, int64_t pixels, int C, dct_stream stream);
int dct_jsd_map_bwd(const float* const* probs, int S, const float* dmap, float* const* dprobs,
                    int64_t pixels, int C, dct_stream stream);
int dct_kl_map_fwd(const float* p, const float* y, float* map, int64_t pixels, int C, float eps,
                   dct_stream stream);
int dct_kl_map_bwd(const float* p, const float* y, const float* dmap, float* dp, int64_t pixels,
                   int C, float eps, dct_stream stream);
/* fused-from-logits fast path used by the step: out[0] = mean JSD over pixels */
int dct_jsd_logits_fwd(const float* const* logits, int S, int64_t pixels, int C, float* out1,
                       void* workspace, size_t workspace_bytes, dct_stream stream);
int dct_jsd_logits_bwd(const float* const* logits, int S, int64_t pixels, int C, const float* gscale,
                       float gmul, float* const* dlogits, int accumulate, dct_stream stream);
/* The step's whole JSD section in one pass (round 5): out1[0] = mean JSD (dct_jsd_logits_fwd's value, bit for bit), probs[s] (nullable array /
 * entries) = softmax of logits[s] (dct_softmax_fwd), dlogits[s] (nullable array: no gradients) (=|+=) the gradient dct_jsd_logits_bwd writes.
 * Two launches instead of five between the join of the S forward passes and the fork of the S backward passes (cotraining_totalloss.py:219-227). */
int dct_jsd_logits_step(const float* const* logits, int S, int64_t pixels, int C, float* out1, float* const* probs,
                        const float* gscale, float gmul, float* const* dlogits, int accumulate,
                        void* workspace, size_t workspace_bytes, dct_stream stream);
/* out[0] = mean_pix sum_c y*(log(y+eps)-log(p+eps)), p = softmax(p_logits), y = softmax(y_logits) */
int dct_kl_logits_fwd(const float* p_logits, const float* y_logits, int64_t pixels, int C, float eps,
                      float* out1, void* workspace, size_t workspace_bytes, dct_stream stream);
int dct_kl_logits_bwd(const float* p_logits, const float* y_logits, int64_t pixels, int C, float eps,
                      const float* gscale, float gmul, float* dp_logits, int accumulate, dct_stream stream);
/* cls[pix] = argmax_c x[pix][c] (first max), int64  (AEGenerator.py:25 pseudo-labels; Dice one-hot) */
int dct_argmax(const float* x, int64_t* cls, int64_t pixels, int C, dct_stream stream);

/* ---- K11: FGSM tail  x_adv = x + eps*sign(g), noise = eps*sign(g) (AEGenerator.py:35-51) ---- */
int dct_fgsm_step(const float* x, const float* g, float eps, float* x_adv, float* noise, int64_t n,
                  dct_stream stream);

/* ---- K12: Adam over one flat fp32 buffer (torch.optim.Adam, segmentators.py:41; step :248) --
 * g' = grad_scale*g + wd*p; m = m + (g'-m)*(1-b1); v = v*b2 + (1-b2)*g'^2;   (grad_scale: 1, or the inverse of the
 * power-of-two loss-gradient scale of an fp16 step)
 * p -= step_size * m / (sqrt(v)/bc2_sqrt + eps)     (host passes step_size = lr/bc1, bc2_sqrt;
 * betas are doubles so that 1-beta is formed in double and then rounded, as torch does)
 * bf16_shadow (nullable): also writes the updated p rounded to bf16 at the same index. */
int dct_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float step_size,
                  float bc2_sqrt, double beta1, double beta2, float eps, float weight_decay, float grad_scale,
                  void* bf16_shadow, dct_stream stream);
/* Same update with the step-dependent scalars in device memory: state = {step count t, learning rate,
 * table base, table length} (four doubles), table (nullable) = {1 - beta1^t, sqrt(1 - beta2^t)} double pairs the
 * host computed for steps t = base+1 .. base+length.  Stream-ordered: t += 1, then the pair for step t is read from
 * the table (bit-identical to dct_adam_flat with the same host scalars); outside the table it is formed on
 * the device in double: step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t).  No per-step host
 * scalar, so the launch can be replayed from a captured HIP graph (torch.optim.Adam's `capturable` mode is
 * the reference-side analogue). */
int dct_adam_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, double* state,
                      const double* table, double beta1, double beta2, float eps, float weight_decay, float grad_scale,
                      void* bf16_shadow, dct_stream stream);

/* ---- K2/K3/K4/K6/K7/K8: Enet layers (arch/enet.py:8-243) -------------------------------------
 * Enet's widths are 1..128 channels (internal 3/16/32): HBM- and launch-bound, so these are direct
 * fused kernels, not GEMMs.  Every consumer reads its input through the producer's BatchNorm +
 * activation ("normalise on load"): in = act(scale[c]*raw + shift[c]); mode 0 none, 1 affine,
 * 2 affine+PReLU(slope[c]), 3 affine+ReLU.  All per-channel vectors are fp32 device arrays. */
typedef struct dct_enet_tf {
  const float* scale; const float* shift; const float* slope;
  int32_t mode;
} dct_enet_tf;

/* Every Enet entry point takes `dtype` (the storage type T of the bf16/f32 activations) and an
 * `f32_mask`: bit k set = the k-th dct_view argument of that call (in declaration order) is stored
 * in fp32 regardless of `dtype`.  In bf16 mode the RAW conv outputs stay fp32 (BatchNorm subtracts
 * their mean, which would cancel most of bf16's 8 mantissa bits); block outputs and gradients are T. */

/* Generic small-channel convolution (<= 128 channels each side).  Replaces nn.Conv2d /
 * nn.ConvTranspose2d of enet.py:21,52-109 and both of their data gradients:
 *   direct:      y[oy,ox,o] = sum W(o,r,s,i) * in[oy*stride - pad + r*dil, ..., i]
 *   transposed:  y[oy,ox,o] = sum W(o,r,s,i) * in[(oy + pad - r*dil)/stride, ..., i]   (where divisible)
 * with W(o,r,s,i) = w[o*ws_out + (r*S+s)*ws_tap + i*ws_in] (fp32 master weights, any role).
 * Epilogue: + bias[o] -> + resid_grad*[resid_mask > 0] (residual-branch gradient of a bottleneck)
 * -> (+= y when d->accumulate).  f32_mask bits: 0 x, 1 y, 2 resid_grad, 3 resid_mask. */
int dct_enet_conv(const dct_view* x, const float* w, const float* bias, const dct_enet_tf* tf,
                  const dct_view* y, const dct_conv_desc* d, int transposed,
                  int ws_out, int ws_tap, int ws_in,
                  const dct_view* resid_grad, const dct_view* resid_mask,
                  int f32_mask, int dtype, dct_stream stream);
/* The same convolution (no residual gate) with the consumer BatchNorm's batch statistics riding along: where the MFMA form is
 * taken (bf16 / f16, >= 16 input channels) and the output has <= stats_capacity_rows tiles of 32 pixels, every tile writes
 * its per-channel {sum, sum of squares, 0} as doubles to stats_partial[tile][y.c][3] and *stats_rows = the tile count;
 * dct_enet_bn_fwd_stats_rows(..., workspace = stats_partial, partial_rows = *stats_rows) then only folds them.  Otherwise
 * *stats_rows = 0 and the statistics need the usual reduction. */
int dct_enet_conv_stats(const dct_view* x, const float* w, const float* bias, const dct_enet_tf* tf,
                        const dct_view* y, const dct_conv_desc* d, int transposed,
                        int ws_out, int ws_tap, int ws_in, int f32_mask, int dtype,
                        double* stats_partial, int stats_capacity_rows, int* stats_rows, dct_stream stream);

/* nn.BatchNorm2d bookkeeping for all layers of a network in one launch: layer k (record k of records_dev)
 *   { float* running_mean; float* running_var; int64* num_batches_tracked (nullable); int32 c, mean_off, var_off, pad; }  (40 bytes)
 * gets r <- (1 - momentum) r + momentum * stats[off + i] for both running statistics (stats: the flat buffer a forward pass
 * left its batch means / UNBIASED variances in) and num_batches_tracked += 1 -- what F.batch_norm does per call in the
 * reference (arch/enet.py:22,55-122).  Replaces four torch._foreach launches per forward pass. */
int dct_bn_running_update(const void* records_dev, int n_layers, const float* stats, float momentum, dct_stream stream);
/* out = a + b (+ c if non-NULL) over n floats (n % 4 == 0, 16-byte aligned): the per-pass flat gradient buffers of one model,
 * added in pass order -- ((labeled + unlabeled) + adversarial), bit for bit the in-place accumulation of sequential passes. */
int dct_flat_sum(float* out, const float* a, const float* b, const float* c, long long n, dct_stream stream);
/* x *= scale over n floats (any 4-byte aligned pointer, any n >= 1): the 1/world of a gradient average after a SUM all-reduce
 * (ddp.py::FlatGradSync, for models whose optimizer does not fold the factor into its update). */
int dct_flat_scale(float* x, float scale, long long n, dct_stream stream);

size_t dct_enet_reduce_workspace_bytes(int channels);
/* nn.BatchNorm2d(eps 1e-3, momentum 0.1) forward statistics of a raw conv output (enet.py:22,55-122):
 * training: batch mean / biased var (double accumulation, fixed-order fold), running stats updated
 * with the unbiased var; eval: running stats.  Writes scale = gamma*invstd, shift = beta - mean*scale
 * (what consumers apply on load) and save_mean / save_invstd for the backward.  f32_mask bit 0: raw.
 * Deferred running statistics: with running_mean = running_var = NULL in training mode nothing is updated and save_var
 * (nullable) receives the batch's UNBIASED variance, so that the caller can apply r = (1 - momentum) r + momentum b for
 * several forward passes later, in the reference's order (the passes themselves may then run concurrently). */
int dct_enet_bn_fwd_stats(const dct_view* raw, const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, int training,
                          float* scale, float* shift, float* save_mean, float* save_invstd, float* save_var,
                          int f32_mask, int dtype, void* workspace, size_t workspace_bytes, dct_stream stream);
int dct_enet_bn_fwd_stats_rows(const dct_view* raw, const float* gamma, const float* beta, float eps, float momentum,
                               float* running_mean, float* running_var, int training,
                               float* scale, float* shift, float* save_mean, float* save_invstd, float* save_var,
                               int f32_mask, int dtype, void* workspace, size_t workspace_bytes, int partial_rows, dct_stream stream);
/* Backward of act(BN(raw)) given g = grad wrt the activation output (optionally gated by
 * g_mask > 0, the ReLU of the bottleneck sum): dgamma/dbeta/dslope += ..., and
 * draw = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)) (training; eval mode: gamma*invstd*dz with the
 * running statistics).  c1c2: 2*C floats of scratch.  f32_mask bits: 0 raw, 1 g, 2 g_mask, 3 draw. */
int dct_enet_bn_bwd(const dct_view* raw, const dct_view* g, const dct_view* g_mask,
                    const float* scale, const float* shift, const float* slope, int act,
                    const float* mean, const float* invstd,
                    float* dgamma, float* dbeta, float* dslope, float* c1c2, int training,
                    const dct_view* draw, int f32_mask, int dtype, void* workspace, size_t workspace_bytes,
                    dct_stream stream);
int dct_enet_bn_bwd_rows(const dct_view* raw, const dct_view* g, const dct_view* g_mask,
                         const float* scale, const float* shift, const float* slope, int act,
                         const float* mean, const float* invstd,
                         float* dgamma, float* dbeta, float* dslope, float* c1c2, int training,
                         const dct_view* draw, int f32_mask, int dtype, void* workspace, size_t workspace_bytes,
                         int partial_rows, dct_stream stream);
/* The data-gradient convolution y = dgrad(x) whose output is the gradient wrt act(BN(bn_raw)) of the producing layer, with that
 * BatchNorm's backward sums riding along (MFMA form only, as dct_enet_conv_stats): every tile of 32 pixels writes
 * {sum dz, sum dz xhat, sum g z [z<0]} per channel to stats_partial[tile][y.c][3]; dct_enet_bn_bwd_rows(..., workspace =
 * stats_partial, partial_rows = *stats_rows) then skips its reduction launch.  *stats_rows = 0: nothing written. */
int dct_enet_conv_bnbwd_stats(const dct_view* x, const float* w, const dct_view* y, const dct_conv_desc* d, int transposed,
                              int ws_out, int ws_tap, int ws_in, int f32_mask, int dtype,
                              const dct_view* bn_raw, const float* bn_scale, const float* bn_shift, const float* bn_slope, int bn_act,
                              const float* bn_mean, const float* bn_invstd,
                              double* stats_partial, int stats_capacity_rows, int* stats_rows, dct_stream stream);
/* out[c] += sum over pixels of x[.., c]   (bias gradients).  f32_mask bit 0: x. */
int dct_enet_channel_sum(const dct_view* x, float* out, int f32_mask, int dtype, void* workspace, size_t workspace_bytes,
                         dct_stream stream);
/* Bottleneck tail out = relu(main + act(bn(raw))) (enet.py:132-149), mode:
 *   0 regular: main = main_in;   1 down: main = maxpool2x2(main_in) zero-padded to out.c channels,
 *   argmax codes written to idx [N,h,w,idx_channels];   2 up: main = max_unpool(bn(rawm)) by idx;
 *   3 initial block (enet.py:26-30): out = cat(prelu(bn(raw)), maxpool2x2(main_in = the image)).
 * f32_mask bits: 0 raw, 1 main_in, 2 rawm, 3 out. */
int dct_enet_tail_fwd(const dct_view* raw, const dct_enet_tf* tf, const dct_view* main_in,
                      const dct_view* rawm, const dct_enet_tf* tfm, uint8_t* idx, int idx_channels,
                      int mode, const dct_view* out, int f32_mask, int dtype, dct_stream stream);
/* Main-branch gradient of those tails: 1 down -> dst = routed grad at double resolution;
 * 2 up -> dst = grad of bn(rawm) (gather); 3 initial -> dst (image grad) (+)= pooled-channel grad
 * (out_mask = the input image).  f32_mask bits: 0 dout, 1 out_mask, 2 dst. */
int dct_enet_tail_bwd(const dct_view* dout, const dct_view* out_mask, const uint8_t* idx, int idx_channels,
                      int mode, int accumulate, const dct_view* dst, int f32_mask, int dtype, dct_stream stream);
/* dw[a.c][R][S][b.c] += sum_pixels tfa(a[p]) (x) tfb(b[p*stride - pad + tap*dil]).
 * Conv2d: a = draw, b = layer input; ConvTranspose2d: a = layer input, b = dy.  f32_mask bits: 0 a, 1 b. */
size_t dct_enet_wgrad_workspace_bytes(const dct_view* a, const dct_view* b, const dct_conv_desc* d);
int dct_enet_wgrad(const dct_view* a, const dct_enet_tf* tfa, const dct_view* b, const dct_enet_tf* tfb,
                   float* dw, const dct_conv_desc* d, int f32_mask, int dtype,
                   void* workspace, size_t workspace_bytes, dct_stream stream);

/* ---- K6 for the BatchNorm'd UNet: nn.BatchNorm2d(C) + ReLU (arch/network.py:138-146,181-183,243-290) -------------------
 * C a power of two, 8 <= C <= 2048; dense-channel NHWC views with 16-byte aligned rows.  Forward: batch statistics of `raw`
 * (training; double accumulation, fixed-order fold; running statistics updated with the unbiased variance) or the running
 * statistics (eval) -> scale = gamma * invstd, shift = beta - mean * scale, save_mean / save_invstd (nullable), and
 * y = relu?(scale * raw + shift) when y != NULL.  Backward: g = gradient wrt y;
 * dgamma / dbeta (= | +=, nullable) and draw = scale * (dz - mean(dz) - xhat * mean(dz * xhat)), dz = g * [y > 0] when relu
 * (eval mode: draw = scale * dz).  c1c2: 2 * C floats of scratch. */
size_t dct_bn_workspace_bytes(int channels);
int dct_bn_fwd(const dct_view* raw, const float* gamma, const float* beta, float eps, float momentum,
               float* running_mean, float* running_var, int training,
               float* scale, float* shift, float* save_mean, float* save_invstd,
               const dct_view* y, int relu, int dtype, void* workspace, size_t workspace_bytes, dct_stream stream);
int dct_bn_bwd(const dct_view* raw, const dct_view* g, const float* scale, const float* shift,
               const float* mean, const float* invstd, float* dgamma, float* dbeta, int accumulate,
               float* c1c2, int training, int relu, const dct_view* draw, int dtype,
               void* workspace, size_t workspace_bytes, dct_stream stream);

/* ---- Dice accumulation on device (metrics/dice_meter.py:12-83) ----------------------------
 * inter[b][c], psum[b][c], gsum[b][c] (int32, zeroed by caller) from logits argmax vs gt. */
int dct_dice_counts(const float* logits, const int64_t* gt, int B, int64_t pixels_per_image, int C,
                    int32_t* inter, int32_t* psum, int32_t* gsum, dct_stream stream);

/* Dice rows of one DiceMeter.add from those counts (2-D: one row per slice; 3-D: one row for the batch) and the meter's
 * running sums in the same launch: dice[rows][C] fp32; acc (double [2][C+1], caller-zeroed once) += value, value^2 per class and,
 * in column C, for the row mean over the report axes (bit c of axes_mask).  B <= 64, C <= 8. */
int dct_dice_update(const int32_t* inter, const int32_t* psum, const int32_t* gsum, int B, int C, int method3d,
                    uint32_t axes_mask, float smooth, float* dice, double* acc, dct_stream stream);

/* ---- per-kernel-class timing (bench.py roofline leg) --------------------------------------
 * When enabled every launch made through this library is bracketed by hipEvents on its stream;
 * dct_prof_read synchronises and returns accumulated milliseconds and launch counts per class. */
enum { DCT_PROF_IGEMM = 0, DCT_PROF_WGRAD = 1, DCT_PROF_POINTWISE = 2, DCT_PROF_LOSS = 3,
       DCT_PROF_ADAM = 4, DCT_PROF_OTHER = 5, DCT_PROF_NCLASS = 6 };
int dct_prof_enable(int on);
/* Process-wide tuning / A-B knobs for benchmarking (never needed for correctness; defaults are the
 * shipped configuration). */
enum { DCT_TUNE_IGEMM_SPLIT = 1,   /* >= 1: force the split-K factor; -1 (default): planner's choice */
       DCT_TUNE_WGRAD_CHUNKS = 3,  /* >= 1: force the number of pixel chunks (split-K) of wgrad */
       DCT_TUNE_IGEMM_HALO = 7,    /* 1 (default): shared-halo kernel for 3x3 stride-1 layers on large images; 0: per-tap kernel everywhere
                                      (the reference path of the bit-level cross-checks in tests/test_kernels_gpu.py) */
       DCT_TUNE_WGRAD_ROWS = 8,    /* 1 (default): filter-row weight-gradient kernel (three taps share the x strip); 0: per-tap kernel */
       DCT_TUNE_WGRAD_ROWS_FILL = 9,   /* percent (default 70): minimum fill of the filter-row kernel's 64-pixel K-steps */
       DCT_TUNE_IGEMM_PACKED = 10,     /* 1 (default): packed-rows shared-halo kernel for 3x3 stride-1 layers on small images; 0: per-tap kernel */
       DCT_TUNE_ENET_WGRAD_BLOCKS = 12,/* 1..1024 (default 1024): cap on the pixel chunks (blocks) of dct_enet_wgrad */
       DCT_TUNE_WGRAD_TARGET = 14,     /* >= 64 (default: see wgrad.hip): block target of the per-tap weight-gradient kernel */
       DCT_TUNE_WGRAD3_TARGET = 15,    /* >= 64: block target (4-wave units) of the filter-row weight-gradient kernel */
       DCT_TUNE_IGEMM_SPLIT_TARGET = 16, /* >= 64 (default 450): block target of a split-K conv layer */
       DCT_TUNE_IGEMM_HALO_MIN_BLOCKS = 19,  /* default 400: fewest blocks for which the shared-halo patch kernel is taken */
       DCT_TUNE_IGEMM_HALO_COVER = 20,       /* percent (default 75): least image cover of its 8 x 16 patches */
       DCT_TUNE_IGEMM_PACKED_SPLIT = 23,     /* default 160: packed-rows kernel splits layers with fewer blocks over channel slices */
       DCT_TUNE_IGEMM_PACKED_FILL = 24,      /* percent (default 50; 76 until round 5): least fill of the packed-rows kernel's 128-pixel tiles */
       DCT_TUNE_ENET_MFMA = 26,              /* bf16 / f16 Enet: bit 0 = MFMA form of the convolutions with >= 16 input channels,
                                                bit 1 = of the weight gradients; 3 (default), 0 = the fp32 VALU kernels */
       DCT_TUNE_ENET_MWGRAD_WAVES = 28,      /* >= 64 (default 2048): waves an MFMA weight-gradient launch aims for */
       DCT_TUNE_IGEMM_XCD = 39,              /* 1 (default): the per-tap and packed-rows conv kernels deal the blocks of a weights-heavy layer XCD by XCD
                                                (each weight is fetched into ONE L2 slice); 0: natural 3-D grids; 2: on every layer */
       DCT_TUNE_LEAN = 38 };                 /* bit mask (default 31: all set) of the instruction-lean loop forms, each bit-identical to
                                                the plain form it replaces (0 = the plain forms, the tests' reference): bit 0 = filter-row weight gradient,
                                                bit 1 = packed-rows conv kernel, bit 2 = per-tap weight gradient, bit 3 = per-tap conv kernel, bit 4 (with bit 0, round 5) =
                                                the filter-row loop skips the 16-row sub-steps of a K-step that hold no dy pixel (exact zeros) */
/* (Knob numbers are stable across rounds; the gaps are A/B switches of variants that were measured slower and removed with their
 *  kernels -- DESIGN.md 4.1 / 4.2: register-staged bf16 kernels, 4-wave tiles, scattered epilogue stores, the 32x32x16 shared-halo
 *  form, XCD-aware tile orders, the weight-ring and four-fat-wave shared-halo tiles, one / four wave groups and per-tap reads in the
 *  filter-row weight gradient, the scalar Enet reductions, the channel-owner BatchNorm, the vector BatchNorm-backward apply, the Enet finalizes riding in their
 *  producers' last blocks (34), the one-block-per-CU ping-pong conv tile (35, 37) -- and planner constants whose
 *  sweeps are on file: profiles/r03_knob_sweeps.txt.) */
int dct_tune_set(int knob, int value);
int dct_prof_read(double* ms_per_class, int64_t* launches_per_class, int reset);
/* Shader-clock probe: ONE wave on `stream` that sleeps until ref_ticks ticks of the constant 100 MHz reference counter have passed
 * and leaves out2[0] = shader-clock cycles (s_memtime), out2[1] = reference ticks (s_memrealtime) of its life: the clock the chip
 * held meanwhile is 0.1 GHz * out2[0] / out2[1].  Launched on a stream of its own beside the captured step, it samples the clock
 * UNDER that load (bench.py `roofline.shader_clock_ghz_during_the_step`).  out2: two device uint64.  ref_ticks <= 1e8 (one second). */
int dct_clock_probe(unsigned long long* out2, unsigned long long ref_ticks, dct_stream stream);
/* Phase stamp: a one-thread launch on `stream` that appends the 100 MHz reference counter (s_memrealtime) to a ring: slot[0] counts the site's
 * stamps, slot[1 + n % ring] holds the n-th (device uint64[1 + ring], zeroed by the caller).  Diagnostic: queued between the phases of a step, it
 * dates them in the real, pipelined schedule of the replayed graph (CoTrainer.phase_stamps, tools/phase_stamps.py). */
int dct_stamp(unsigned long long* slot, unsigned ring, dct_stream stream);

/* "De-normalise on load": a data-gradient convolution whose input is the BatchNorm-backward result of the layer in front of it
 * computes that result where it would load it -- raw = that layer's fp32 output, tf = its scale / shift / slope (mode != 0),
 * bin = {gradient wrt its activation g, optional ReLU mask of g, saved mean / invstd, c1c2 from dct_enet_bn_bwd_sums} -- so the
 * elementwise apply launch is no longer on the chain between the sums and this convolution (arch/enet.py: it is only launched,
 * as a leaf, where a weight gradient needs the tensor).  Same arithmetic as the apply kernel (one shared definition), rounded to
 * the compute type as the stored tensor would be.  Optional residual gate / accumulate as dct_enet_conv; optional output-side
 * BatchNorm-backward sums as dct_enet_conv_bnbwd_stats (bn_raw != NULL).  MFMA form only (bf16 / f16, >= 16 channels, 16-byte
 * aligned views): DCT_ERR_UNSUPPORTED otherwise -- the caller then materialises the tensor (dct_enet_bn_bwd_apply) and calls
 * dct_enet_conv. */
typedef struct dct_enet_bwd_in {
  const dct_view* g; const dct_view* g_mask;
  const float* mean; const float* invstd; const float* c1c2;
} dct_enet_bwd_in;
int dct_enet_conv_bwd_in(const dct_view* raw, const float* w, const dct_enet_tf* tf, const dct_enet_bwd_in* bin,
                         const dct_view* y, const dct_conv_desc* d, int transposed, int ws_out, int ws_tap, int ws_in,
                         const dct_view* resid_grad, const dct_view* resid_mask, int f32_mask, int dtype,
                         const dct_view* bn_raw, const float* bn_scale, const float* bn_shift, const float* bn_slope, int bn_act,
                         const float* bn_mean, const float* bn_invstd,
                         double* stats_partial, int stats_capacity_rows, int* stats_rows, dct_stream stream);
/* The two halves of dct_enet_bn_bwd[_rows] on their own: the sums (reduction, or the rows a convolution's epilogue wrote, + finalize:
 * parameter gradients and c1c2) and the elementwise apply (draw from raw, g, c1c2).  leaf != 0: the apply's result is only read by
 * weight gradients, so under dct_leaves_begin it is held back with them. */
int dct_enet_bn_bwd_sums(const dct_view* raw, const dct_view* g, const dct_view* g_mask,
                         const float* scale, const float* shift, const float* slope, int act,
                         const float* mean, const float* invstd,
                         float* dgamma, float* dbeta, float* dslope, float* c1c2, int training,
                         int f32_mask, int dtype, void* workspace, size_t workspace_bytes, int partial_rows, dct_stream stream);
int dct_enet_bn_bwd_apply(const dct_view* raw, const dct_view* g, const dct_view* g_mask,
                          const float* scale, const float* shift, const float* slope, int act,
                          const float* mean, const float* invstd, const float* c1c2,
                          const dct_view* draw, int f32_mask, int dtype, int leaf, dct_stream stream);

/* ---- grouped passes -------------------------------------------------------------------------------------------
 * Independent passes with identical shapes (the 2S forward, then the 2S backward passes of a co-training step:
 * cotraining_totalloss.py:208-227,247) as ONE chain of launches: between dct_group_begin(members) and dct_group_end the
 * dct_enet_* entry points record their launches for the member selected by dct_group_member(m) instead of issuing them
 * (every other entry point still launches at once: do not call them inside a group); dct_group_end(stream, ...) issues
 * entry k of all members as one launch (blockIdx.z = member) where kernel, grid, block and LDS bytes agree, else member by
 * member.  Same kernel bodies on the same operands: bit-identical to the passes launched one after the other.  The
 * operands of every recorded launch must stay allocated until dct_group_end returns.  members <= dct_group_max(). */
int dct_group_begin(int members);
int dct_group_member(int member);
int dct_group_max(void);
int dct_group_abort(void);
int dct_group_end(dct_stream stream, int* grouped_launches, int* single_launches);
/* Leaves of a backward pass on another queue: from dct_leaves_begin on, the launches of dct_enet_wgrad and dct_enet_channel_sum
 * (weight / bias gradients: nothing later in the pass reads them) are held back while all other entry points launch as usual;
 * dct_leaves_flush(stream, &n) issues what is held so far on `stream` and keeps recording; dct_leaves_end issues the rest.
 * The caller orders `stream` after the producing chain (an event before the flush) and joins it before the gradients are read;
 * operands stay allocated until then.  Not nestable with a group. */
int dct_leaves_begin(void);
int dct_leaves_flush(dct_stream stream, int* launches);
int dct_leaves_end(dct_stream stream, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* DCT_H_ */
