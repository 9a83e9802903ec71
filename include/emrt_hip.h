/* emrt_hip.h -- C-ABI of libemrt_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the EMRT per-tile
 * forward/backward hot path (ResNet-50 -> MSDeformAttn encoder -> cross-attention decoder -> segmentation head).
 *
 * The reference (peach-xiao/EMRT) is pure PaddlePaddle Python and has no FFI layer of its own: the
 * arithmetic lives in Paddle's cuDNN/cuBLAS/CUDA operators.  Each entry point below therefore names the Paddle
 * operator CALL SITE in the reference that it replaces (paths relative to
 * /root/reference/semantic_segmentation/src/models/).  The reference-side binding a maintainer would add is the
 * ctypes stub shown in INTEGRATION.md (the same one emrt_amd/_lib.py generates from this header).
 *
 * Conventions (all entry points):
 *  - plain C, no C++/torch types; every pointer is a caller-owned DEVICE pointer unless marked "host";
 *  - the library never allocates, frees or synchronises: work is enqueued on `stream` (a hipStream_t; 0 = default),
 *    so calls can be captured in a hipGraph; workspaces are sized by the *_workspace_bytes queries;
 *  - returns 0 on success, negative on error; emrt_last_error() gives a thread-local message;
 *  - dtype: 0 = float32, 1 = bfloat16, 2 = float16 (storage type of activations / packed weights; accumulation is always
 *    fp32).  float16 is inference-only (the sliding-window configuration, src/api/infer.py:22-80): the forward entry points
 *    take it, every backward / training entry point returns an error for it;
 *  - activations are NHWC: `ld` = pixel (row) stride in elements, `bs` = batch stride in elements, so a level slab
 *    of the [B, Lv, C] token tensor or a channel slice of a concat buffer is addressed in place.
 */
#ifndef EMRT_HIP_H
#define EMRT_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMRT_DTYPE_F32 0
#define EMRT_DTYPE_BF16 1
#define EMRT_DTYPE_F16 2

const char* emrt_last_error(void);
int emrt_abi_version(void);
int emrt_device_info(int* cu_count, size_t* lds_bytes, char* arch, int arch_len);
/* Developer / test knobs of the dispatchers (forced tile shapes, kernel variants).  The table is filled ONCE from the
 * environment (EMRT_<NAME>) when the library is loaded; no entry point calls getenv() afterwards.  name (HOST string) is one of:
 * conv_tile, wgrad_split, no_ksplit128, ln_bwd_rows, ln_bwd_max_blocks, wgrad_no_overwrite, bn_operand_blocks, no_s2_dgrad, wgroup_blocks, wgroup_min_steps, wgroup_max, thin_cblk, thin_blocks, thin_ch, no_thin_bwd, pair_max, msda_fwd_global, msda_bwd_global, msda_fwd_chunks,
 * msda_fwd_threads, msda_fwd_probe, wgrad_nst, igemm64_nst, bn_block_kb, ln_atomic, gn_group_blocks, gn_stat_rows, msda_lds_min_pairs, msda_bwd_dref_lds,
 * gn_bwd_stat_rows, gn_apply_rows, xk, mha_valu, msda_scatter_merge, msda_scatter_mfma, msda_mf_bands, sgd_nt, ln_bwd_threads, no_bna, memcpy_kernel,
 * wgroup8, wgroup8_blocks, wgroup8_min_work, mha_bwd_split.
 * Not thread-safe against concurrent launches; production code never calls these. */
int emrt_set_tuning(const char* name, int value);
int emrt_get_tuning(const char* name, int* value);
/* Registers DEVICE memory (256-byte aligned; nullptr, 0 to unregister) that kernels may use for partial sums between two of their own
 * launches on the caller's stream: today the 256x256 weight-gradient kernel's per-block partial tiles (<= 64 MiB: one 256 KiB tile per CU)
 * for train.py:142-149's loss.backward() of the large layers.  Without it that kernel adds its tiles into dW with fp32 atomics.  The
 * memory must stay valid (and the pointer unchanged across hipGraph replays) until it is unregistered.  `stream` (ABI 5) is the ONE stream
 * whose launches may use it: a call on any other stream falls back to the atomic epilogue (partial tiles + reduce launch are only ordered
 * within a stream).  ABI 7: a region of more than 24 MiB (a multiple of 256 bytes) gives its LAST 8 MiB + 64 KiB to the partial tiles and arrival counters of the
 * convolutions' cross-block K split (emrt_conv2d / emrt_conv2d_bwd on few-tile, long-K layers: partial fp32 tiles of S blocks per output tile
 * go through the region, the last block to arrive sums them and runs the epilogue -- nn.Conv2D of the 8x8 / 16x16 ResNet stages,
 * paddle_vision_resnet.py:111-119); the counters are zeroed here with a memset enqueued on `stream` and every launch leaves them zero.  Without
 * a registered region (fp32 contexts) those layers run the in-block K split. */
int emrt_set_scratch(void* ptr, size_t bytes, void* stream);

/* ---- convolution / linear as implicit GEMM (MFMA 32x32) ---------------------------------------------------
 * replaces nn.Conv2D: backbones/paddle_vision_resnet.py:108-123,192-198,226-233; paddle_EMRT.py:16-23,63,85-91,
 * 134-138,201-209; decoders/fcn_head.py:52,66; EMRT_utils/transformer_encoder_decoder.py:125-144,374-378
 * and nn.Linear / F.linear: transformer_encoder_decoder.py:36-42,118-121,259-262,371; EMRT_utils/layers.py:221-229,306.
 * dilation: spacing of the kernel taps (1 = dense; the dilated stages of the resnet50c backbone, backbones/resnet.py:65-66,112-124).
 * out_scale (nullable, fp32 [OC]): the accumulator is multiplied by it before the bias -- an eval-mode BatchNorm folded into the
 * convolution (emrt_bn_fold gives scale and shift; the shift goes in as `bias`), so inference launches no BatchNorm kernel.
 * mode 0: out = conv(in, W) * out_scale [+bias][+residual][relu];  w_packed = [OC][KH][KW][C]  (dims N,H,W,C describe `in`)
 * mode 1: data gradient; `in` is dY (N,H,W,C = its dims), out is dX (OH,OW,OC), w_packed = [Cin][KH][KW][Cout].
 * bn_stats (nullable): fp64 [8][2*OC] (8 replicas, see BatchNorm below), pre-zeroed; the epilogue adds per-channel sum / sum-of-squares of the stored outputs
 * (the BatchNorm statistics of the layer that follows, fused so the activation is not re-read). */
int emrt_conv2d(const void* in, const void* w_packed, void* out, const float* bias, const void* residual, int N, int H, int W, int C, int ldin, long long in_bs, int OH, int OW, int OC, int ldout, long long out_bs, int ldres, long long res_bs, int KH, int KW, int stride, int pad, int mode, int relu, int out_f32, double* bn_stats, const void* mask_y, int ldy, long long y_bs, int dilation, const float* out_scale, int dtype, void* stream);
/* out [M][OC] = dropout_p(relu(in [M][C] . w_packed^T + bias)), the mask drawn in the GEMM epilogue: replaces nn.Linear -> F.relu -> nn.Dropout of
 * the transformer FFN (EMRT_utils/transformer_encoder_decoder.py:118-121,157-161 encoder; :259-262,276-280 decoder) in one launch.  Row strides
 * ldin / ldout in elements; 0 < p < 1; seed: the device seed word (emrt_counter_add advances it per step); salt: per call site; OC % 8 == 0;
 * dtype 0 / 1.  Kept values are scaled by 1 / (1 - p).  Backward: emrt_conv2d_bwd of the CONSUMER with mask_y = out, mask_scale = 1 / (1 - p)
 * (a stored value > 0 <=> kept and past the ReLU), then emrt_conv2d_bwd of this layer on the masked gradient: no mask tensor, no seed. */
int emrt_conv2d_drop(const void* in, const void* w_packed, void* out, const float* bias, int M, int C, int ldin, int OC, int ldout, float p, const unsigned long long* seed, unsigned salt, int dtype, void* stream);

/* ABI 8: out = conv([relu](BatchNorm_train(in))) with the BatchNorm applied by the convolution's own operand loads (`in` is the RAW output of the producing
 * conv; the normalised map is written to a_out, dense [N][H][W][C], on the way) -- emrt_bn_apply + emrt_conv2d in one launch for the BatchNorm -> ReLU -> conv
 * chains of paddle_vision_resnet.py:129-149 (bn1 -> relu -> conv2, bn2 -> relu -> conv3), paddle_EMRT.py:16-23 (Conv2dBlock) and :201-209 (cls_psp).  Convolution
 * arguments as emrt_conv2d (forward), BatchNorm arguments as emrt_bn_apply (sums complete; mean / invstd saved, running statistics updated).  1x1 / 3x3 "same"
 * stride-1 layers on the 64x64-tile kernels only: emrt_conv2d_bna_supported (same arguments; no launch, no error state) says whether this layer is one, and
 * the caller keeps the two-launch form otherwise.  Bit-identical to the two-launch form (out AND a_out). */
int emrt_conv2d_bna_supported(const void* in, const void* w_packed, void* out, const float* bias, const void* residual, int N, int H, int W, int C, int ldin, long long in_bs, int OH, int OW, int OC, int ldout, long long out_bs, int ldres, long long res_bs, int KH, int KW, int stride, int pad, int relu, int out_f32, double* bn_stats, int dilation, const double* sums, double count, float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma, const float* beta, int in_relu, void* a_out, int dtype, void* stream);
int emrt_conv2d_bna(const void* in, const void* w_packed, void* out, const float* bias, const void* residual, int N, int H, int W, int C, int ldin, long long in_bs, int OH, int OW, int OC, int ldout, long long out_bs, int ldres, long long res_bs, int KH, int KW, int stride, int pad, int relu, int out_f32, double* bn_stats, int dilation, const double* sums, double count, float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma, const float* beta, int in_relu, void* a_out, int dtype, void* stream);
/* weight gradient, ACCUMULATED (fp32 atomics) into dw [OC][KH][KW][C]; dbias (nullable, [OC]) += sum over pixels of dy */
int emrt_conv2d_wgrad(const void* x, const void* dy, float* dw, int N, int H, int W, int C, int ldx, long long x_bs, int OH, int OW, int OC, int lddy, long long dy_bs, int KH, int KW, int stride, int pad, float* dbias, int dilation, int dtype, void* stream);
/* ---- grouped launches: up to 6 independent SMALL problems (the per-level 3x3 convs of an encoder layer + its attention projections, ...) as ONE
 * launch; problems that are not small vector-path ones are launched one by one instead.  Descriptors are host arrays. */
typedef struct EmrtConvDesc {            /* the arguments of emrt_conv2d, mode 0 */
  const void* in; const void* w_packed; void* out; const float* bias; const void* residual;
  int N, H, W, C, ldin; long long in_bs;
  int OH, OW, OC, ldout; long long out_bs;
  int ldres; long long res_bs;
  int KH, KW, stride, pad, relu;
  double* bn_stats;
  int out_f32;                           /* ABI 7: 1 = fp32 output (the offsets | logits projection of the deformable attention, t_e_d.py:89-92, grouped
                                            with value_proj of the same layer: two independent linears, one launch) */
} EmrtConvDesc;
typedef struct EmrtConvBwdDesc {         /* the arguments of emrt_conv2d_bwd without the fused BatchNorm sums */
  const void* x; const void* dy; const void* w_bwd_packed; void* dx; int lddx; long long dx_bs; int accumulate; float* dw; float* dbias;
  int N, H, W, C, ldx; long long x_bs; int OH, OW, OC, lddy; long long dy_bs; int KH, KW, stride, pad;
} EmrtConvBwdDesc;
int emrt_conv2d_group(const EmrtConvDesc* descs, int n, int dtype, void* stream);
/* dw == NULL in EVERY descriptor (then dbias must be NULL too): the data gradients only, one grouped launch of dgrad tiles */
int emrt_conv2d_bwd_group(const EmrtConvBwdDesc* descs, int n, int dtype, void* stream);
/* ABI 9: data gradients of 1..6 INDEPENDENT layers WITH the fused epilogues of emrt_conv2d_bwd (the producer's ReLU mask, its BatchNorm backward sums,
 * an addend), argument for argument the data-gradient half of that entry point; small vector-path problems (<= 2048 tiles of 64 x 64 each, <= 4096 in
 * all) run as ONE grouped launch, anything else one launch each.  n == 1 is emrt_conv2d_bwd with dw == NULL.  No two problems may write the same dx.
 * Replaces nothing in the reference: paddle's autograd launches every layer's backward on its own (train.py:142-149). */
typedef struct EmrtConvDgradDesc {
  const void* dy; const void* w_bwd_packed; void* dx; int lddx; long long dx_bs; int accumulate;
  int N, H, W, C, OH, OW, OC, lddy; long long dy_bs; int KH, KW, stride, pad, dilation;
  double* bn_stats; const void* mask_y; int ldy; long long y_bs; float mask_scale; const void* stat_x; int ldsx; long long sx_bs;
  const void* addend; int ldadd; long long add_bs;
} EmrtConvDgradDesc;
int emrt_conv2d_dgrad_multi(const EmrtConvDgradDesc* descs, int n, int dtype, void* stream);
/* ---- batched weight gradients (ABI 5): dW(L) needs only x(L) and dy(L), nothing in loss.backward() (train.py:142-149) waits for it.  A
 * caller may therefore run each layer's DATA gradient alone (emrt_conv2d_bwd / emrt_conv2d_bwd_group with dw == NULL), keep x and dy
 * alive, and hand the weight gradients of many layers (any n >= 1; HOST array) to one call here: small vector-path problems run as grouped
 * launches of up to 24 problems whose pixel reductions are cut into far fewer slices than each would need alone to fill the GPU (less fp32
 * atomic traffic into dW); large layers take the 256x256 LDS-DMA kernel and odd shapes the element-wise path, one launch each.  Same
 * arithmetic as emrt_conv2d_wgrad: dw [OC][KH][KW][C] += , dbias (nullable) += . */
typedef struct EmrtWgradDesc {
  const void* x; const void* dy; float* dw; float* dbias;
  int N, H, W, C, ldx; long long x_bs;
  int OH, OW, OC, lddy; long long dy_bs;
  int KH, KW, stride, pad, dilation;
  int dw_is_zero;      /* 1: the caller vouches that dw (and nothing else) is all zero when this call executes (first contribution after the
                          per-step clear): a problem that runs as ONE pixel slice then stores its tiles instead of adding them with atomics */
} EmrtWgradDesc;
int emrt_conv2d_wgrad_group(const EmrtWgradDesc* descs, int n, int dtype, void* stream);

/* backward of one conv / linear layer in one call (dx NHWC with strides lddx / dx_bs, overwritten or accumulated into; dW += ; dbias += when given;
 * dw == NULL (then dbias == NULL): the data gradient only, see emrt_conv2d_wgrad_group; optional fused
 * BatchNorm-backward sums as in emrt_conv2d): small layers are ONE launch that runs the dgrad and wgrad tiles side by side.
 * With mask_y, dx = mask_y > 0 ? mask_scale * dgrad : 0: mask_scale = 1 for a ReLU, 1/(1-p) when mask_y is the output of
 * dropout(relu(.)) -- the FFN's dropout and ReLU backward cost no pass of their own.  addend (same geometry as dx, own strides):
 * dx = dgrad + addend before the mask, i.e. the gradient contributions other consumers made so far are folded in without
 * an accumulate pass; stat_x: the second statistic becomes sum dx * stat_x (the BatchNorm input) instead of sum dx * mask_y,
 * which is what a relu(BatchNorm(x) + residual) join needs -- its reduction pass then disappears too */
int emrt_conv2d_bwd(const void* x, const void* dy, const void* w_bwd_packed, void* dx, int lddx, long long dx_bs, int accumulate, float* dw, float* dbias, int N, int H, int W, int C, int ldx, long long x_bs, int OH, int OW, int OC, int lddy, long long dy_bs, int KH, int KW, int stride, int pad, double* bn_stats, const void* mask_y, int ldy, long long y_bs, float mask_scale, const void* stat_x, int ldsx, long long sx_bs, const void* addend, int ldadd, long long add_bs, int dilation, int dtype, void* stream);

/* ---- BatchNorm / SyncBatchNorm (train: fp64 sums [from the conv epilogue or emrt_bn_stats] -> [all-reduce of sums across
 * ranks] -> apply; eval: running statistics).  replaces nn.BatchNorm2D / nn.SyncBatchNorm (+ReLU, + residual add):
 * paddle_vision_resnet.py:132-147; paddle_EMRT.py:18,22,64,86-91,131,139-141,203,206; fcn_head.py:53.
 * All `sums` buffers are fp64 [8][2*C] -- producers spread their atomics over 8 replicas, consumers add them -- and must be
 * zeroed by the caller before the producing kernel runs. */
size_t emrt_colreduce_workspace_bytes(long long M, int C);
int emrt_bn_stats(const void* x, int ldx, long long M, int C, double* sums, int dtype, void* stream);
int emrt_bn_apply(const void* x, int ldx, const void* res, int ldres, void* y, int ldy, const double* sums, double count, float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma, const float* beta, long long M, int C, int relu, int dtype, void* stream);
/* ABI 6: out = [relu](BN_train(x) + BN_train(res_raw)) -- the join that ends the first block of a ResNet stage, whose shortcut is
 * conv1x1 -> BatchNorm without ReLU (paddle_vision_resnet.py:132-147, 226-233) -- with the shortcut's BatchNorm applied as its raw conv
 * output is loaded (no separate emrt_bn_apply for it, no normalised shortcut map).  r_*: the shortcut's BatchNorm, same meaning as the
 * main one's arguments; both save mean / invstd and update their running statistics.  Backward: the main BatchNorm as after
 * emrt_bn_apply with a residual; the shortcut's through emrt_bn_bwd_reduce / _dx on (res_raw, masked dy). */
int emrt_bn_apply_join(const void* x, int ldx, const void* res_raw, int ldres, void* y, int ldy, const double* sums, double count, float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma, const float* beta, const double* r_sums, double r_count, float r_eps, float r_momentum, float* r_mean, float* r_invstd, float* r_run_mean, float* r_run_var, const float* r_gamma, const float* r_beta, long long M, int C, int relu, int dtype, void* stream);
/* eval: every BatchNorm of a model as an affine map, one launch.  desc: device int64 [n][7] = (gamma offset, beta offset) into
 * `params`, (running mean offset, running variance offset) into `buffers`, C, offset into `out`, offset in `params` of the bias of
 * the convolution feeding this BatchNorm or -1; writes out[o..o+C) = s = gamma / sqrt(var + eps) and
 * out[o+C..o+2C) = beta + (conv_bias - mean) * s: emrt_conv2d's out_scale / bias. */
int emrt_bn_fold(const float* params, const float* buffers, const long long* desc, int n, float eps, float* out, void* stream);
/* ABI 6: mask_gamma / mask_beta of emrt_bn_bwd_reduce and mask_beta of emrt_bn_bwd_dx (nullable; y must then be NULL): the layer's ReLU
 * output was never written because its consumer applied BatchNorm + ReLU on load (emrt_bn_resize_bilinear_fwd, emrt_bn_maxpool_fwd); the
 * mask is re-derived from x with the forward's own expression, relu'(x * invstd * gamma + (beta - mean * invstd * gamma)). */
int emrt_bn_bwd_reduce(const void* x, int ldx, const void* dy, int lddy, const void* y, int ldy, const float* mean, const float* invstd, long long M, int C, double* sums, const float* mask_gamma, const float* mask_beta, int dtype, void* stream);
int emrt_bn_bwd_dx(const void* x, int ldx, const void* dy, int lddy, const void* y, int ldy, void* dx, int lddx, void* dres, int lddres, const float* mean, const float* invstd, const float* gamma, const double* sums, const double* local_sums, double count, float* dgamma, float* dbeta, long long M, int C, const float* beta_y_moments, int sums_vs_x, const float* mask_beta, int dtype, void* stream);
/* ABI 6: the classifier behind conv -> SyncBatchNorm -> ReLU (paddle_EMRT.py:176-179) with the BatchNorm + ReLU applied by its own loads.
 * fwd: out[N][HW][OC] = bias + relu(BatchNorm_train(x)) . w^T; x the RAW map of the producing conv (row stride ldx, image stride x_bs),
 * w_packed the forward-packed [OC][C] weight of a 1x1 conv / linear, OC <= 8, C in {64, 128, 256}; BatchNorm arguments as emrt_bn_apply.
 * bwd (one pass): da = (dy . w) masked by relu(BN(x)) > 0 (dense [N][HW][C]), dW += dy^T . relu(BN(x)), dbias += sum dy (nullable), and
 * `stats` (nullable, ZEROED fp64 [8][2C]) += (sum da, sum da * relu(BN(x))): the sums emrt_bn_bwd_dx takes with beta_y_moments. */
int emrt_bn_pointwise_fwd(const void* x, int ldx, long long x_bs, const void* w_packed, const float* bias, void* out, int ldo, long long o_bs, int N, int HW, int C, int OC, const double* sums, double count, float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma, const float* beta, int relu, int dtype, void* stream);
int emrt_bn_pointwise_bwd(const void* x, int ldx, long long x_bs, const void* dy, int lddy, long long dy_bs, const void* w_bwd_packed, void* da, int ldda, long long da_bs, float* dw, float* dbias, double* stats, int N, int HW, int C, int OC, const float* mean, const float* invstd, const float* gamma, const float* beta, int dtype, void* stream);
/* per-channel sum accumulated into dbias (bias / embedding gradients) */
int emrt_colsum_acc(const void* x, int ldx, long long rows_per_batch, long long x_bs, long long M, int C, float* dbias, void* workspace, int dtype, void* stream);
/* ABI 8: dst[l][c] += sum over the T (<= 8) tensors xs[t] ([B][Lv][C], dense), the batch and the tokens [level_start[l], + level_count[l]) of level l
 * (<= 4 levels; host arrays): the level embedding's gradient (transformer_encoder_decoder.py:447-448) from every encoder layer's query gradient in one launch. */
int emrt_colsum_levels_multi(const void* const* xs, int T, const int* level_start, const int* level_count, int L, int B, int Lv, int C, float* dst, int dtype, void* stream);

/* ABI 8: training-mode BatchNorm (+ ReLU) of up to 8 SMALL independent problems in one launch per pass: the four pyramid-pooling branches
 * (paddle_EMRT.py:61-66,70-78: conv1x1 -> SyncBatchNorm -> ReLU on 1 / 9 / 36 / 64 pooled tokens per image) instead of one emrt_bn_apply forward and one
 * emrt_bn_bwd_reduce + emrt_bn_bwd_dx backward launch per branch.  x / y / dy / dx: [M][C] rows with their strides; `sums`: forward the COMPLETE fp64
 * (sum x, sum x^2) [8][2C] of the producing conv's epilogue, backward a ZEROED [8][2C]; mean / invstd are written by the forward and read by the backward;
 * dgamma / dbeta are accumulated.  Forward bit-identical to emrt_bn_apply.  One rank only (no statistics all-reduce between the passes). */
typedef struct EmrtBnGroupDesc {
  const void* x; void* y; const void* dy; void* dx; double* sums;
  float* mean; float* invstd; float* run_mean; float* run_var; const float* gamma; const float* beta; float* dgamma; float* dbeta;
  double count; float eps, momentum;
  int M, C, ldx, ldy, lddy, lddx, relu;
  const void* res; int ldres, res_hw; long long res_bs;      /* forward, nullable: y = [relu](BN(x)) + res (paddle_EMRT.py:24-29), row r = pixel r % res_hw of image r / res_hw at
                                                                res + image * res_bs + pixel * ldres; the backward re-derives the ReLU mask from x, y is not read */
} EmrtBnGroupDesc;
int emrt_bn_group_apply(const EmrtBnGroupDesc* descs, int n, int dtype, void* stream);
int emrt_bn_group_bwd(const EmrtBnGroupDesc* descs, int n, int dtype, void* stream);

/* ---- GroupNorm(32) [+ erf-GELU] [+ residual]: transformer_encoder_decoder.py:125-144 (conv branch), :378 (input_proj).
 * workspace: PRE-ZEROED fp64, [N*G*2] for fwd (group sums), [N*C*2] for bwd (per-image channel sums). */
int emrt_groupnorm_fwd(const void* x, int ldx, long long x_bs, const void* res, int ldres, long long res_bs, void* out, int ldout, long long out_bs, const float* gamma, const float* beta, float* mean, float* rstd, double* workspace, int N, int HW, int C, int G, float eps, int gelu, int dtype, void* stream);
int emrt_groupnorm_bwd(const void* x, int ldx, long long x_bs, const void* dy, int lddy, long long dy_bs, void* dx, int lddx, long long dx_bs, const float* gamma, const float* beta, const float* mean, const float* rstd, float* dgamma, float* dbeta, double* workspace, int N, int HW, int C, int G, int gelu, int dtype, void* stream);
/* the same over L <= 4 level slabs (rows [level_start[l], +level_hw[l]), hw <= 4096) of token tensors [N][Lv][C], each with
 * its own gamma / beta, in ONE launch: the per-level conv branch of an encoder layer (transformer_encoder_decoder.py:125-144,
 * 163-182).  gamma / beta / dgamma / dbeta / level_* are HOST arrays of L entries; mean / rstd are [L][N*G]. */
int emrt_groupnorm_levels_fwd(const void* x, int ldx, long long x_bs, const void* res, int ldres, long long res_bs, void* out, int ldout, long long out_bs, const float* const* gamma, const float* const* beta, float* mean, float* rstd, const int* level_start, const int* level_hw, int L, int N, int C, int G, float eps, int gelu, double* stat_ws, int dtype, void* stream);
int emrt_groupnorm_levels_bwd(const void* x, int ldx, long long x_bs, const void* dy, int lddy, long long dy_bs, void* dx, int lddx, long long dx_bs, const float* const* gamma, const float* const* beta, const float* mean, const float* rstd, float* const* dgamma, float* const* dbeta, const int* level_start, const int* level_hw, int L, int N, int C, int G, int gelu, double* stat_ws, int dtype, void* stream);

/* ---- residual add + LayerNorm (+ post add): transformer_encoder_decoder.py:199-203,159-160,285-291,278-279
 * z = a (+ b); out = LN(z) * gamma + beta (+ post).  z, mean, rstd are saved for backward.
 * ABI 8: qpos / qpos_rows / q_out (nullable together): q_out[row] = out[row] + qpos[row % qpos_rows] -- the next attention's query with_pos_embed(out, pos)
 * (transformer_encoder_decoder.py:186,283-289) written beside `out`; emrt_layernorm_bwd's dy2 (nullable, [rows][C]) is that query's gradient, summed with dy
 * as it is loaded (dysum, nullable, receives dy + dy2: the gradient of the forward's `post` addend).  One add launch forward and one accumulate launch
 * backward less per attention. */
int emrt_layernorm_fwd(const void* a, const void* b, const void* post, void* z, void* out, const float* gamma, const float* beta, float* mean, float* rstd, long long rows, int C, float eps, float pdrop, const unsigned long long* seed, unsigned salt, const void* qpos, int qpos_rows, void* q_out, int dtype, void* stream);
size_t emrt_layernorm_bwd_workspace_bytes(long long rows, int C);
/* dz_addend (ABI 7, nullable, [rows][C], must not alias dz): a gradient contribution to the residual input that is already known -- the encoder layer's
 * conv-branch tokens reach the layer input through an identity as well (transformer_encoder_decoder.py:202-203) -- summed into dz here instead of by an add launch;
 * dz_branch does not receive it. */
int emrt_layernorm_bwd(const void* z, const void* dy, void* dz, const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta, long long rows, int C, void* workspace, void* dz_branch, float pdrop, const unsigned long long* seed, unsigned salt, const void* dz_addend, const void* dy2, void* dysum, int dtype, void* stream);

/* ---- multi-scale deformable attention core (fused softmax + sampling locations + bilinear gather + weighted sum)
 * replaces transformer_encoder_decoder.py:89-104 and EMRT_utils/utils.py:64-97 (deformable_attention_core_func).
 * value [B][Lv][M*D] (D = 32); offw fp32 [B*Lq][ldo] = M*L*P*2 offsets then M*L*P logits per row;
 * ref fp32 [B or 1][Lq][ref_L][2] (ref_bs = 0 broadcasts over batch; ref_L = L, or 1 to share one point across levels); shapes_hw: HOST int [L][2] = (H_l, W_l); out [B][Lq][M*D].
 * Arithmetic: softmax, coordinates and corner weights in fp32; with a 16-bit value type the corner weights are then rounded to that type and the
 * weighted sum is a chain of 2-way dot products with fp32 accumulation (forward relative L2 error 2.3e-3 in bf16 = output rounding + weight rounding;
 * fp32 maps keep fp32 weights); 16-bit value tensors must span less than 2 GiB (32-bit buffer offsets). */
int emrt_msda_fwd(const void* value, int ldv, long long v_bs, const float* offw, int ldo, const float* ref, long long ref_bs, int ref_L, void* out, int B, int Lq, int Lv, int M, int D, int L, int P, const int* shapes_hw, int dtype, void* stream);
/* Backward.  doffw [B*Lq][ldo] (fp32, or the compute dtype when doffw_compute_dtype != 0 and dtype is a 2-byte type: what the
 * offsets|logits projection's backward GEMM reads) is overwritten; dref [B][Lq][ref_L][2] (nullable) must be ZEROED by the caller (the
 * LDS-staged gradient kernel adds the heads' contributions with atomics; the global-gather kernel overwrites it).  dvalue: when
 * emrt_msda_bwd_uses_lds(shapes_hw, L) == 1 (every level group's fp32 slab fits in LDS) it is [B][Lv][M*D] in the
 * compute dtype, fully overwritten by an LDS-privatised scatter (needs `workspace` of emrt_msda_bwd_workspace_bytes);
 * otherwise it is fp32, must be zeroed by the caller and is accumulated with global atomics.
 * emrt_msda_bwd_workspace_bytes (ABI 4: takes the level shapes and the dtype of the call it sizes): softmax probabilities, the
 * per-block max |dout| partials and -- for large pyramids, where the small levels' scatter blocks are split by queries -- the
 * integer partial slabs; it makes the same plan emrt_msda_bwd will make for these arguments.  ABI 5: shapes_hw == NULL (or L outside
 * 1..4) is an error (returns 0, emrt_last_error() says why), and emrt_msda_bwd takes the size of the workspace it was given and
 * refuses one that is smaller than this function's answer. */
int emrt_msda_bwd_uses_lds(const int* shapes_hw, int L);
size_t emrt_msda_bwd_workspace_bytes(int B, int Lq, int M, int L, int P, const int* shapes_hw, int dtype);
int emrt_msda_bwd(const void* value, int ldv, long long v_bs, const float* offw, int ldo, const float* ref, long long ref_bs, int ref_L, const void* dout, void* dvalue, void* doffw, int doffw_compute_dtype, float* dref, int B, int Lq, int Lv, int M, int D, int L, int P, const int* shapes_hw, void* workspace, size_t workspace_bytes, int dtype, void* stream);

/* ---- fused softmax(QK^T/sqrt(d)) V with dropout on the weights: EMRT_utils/layers.py:283-303 (L <= 128, D = 32).
 * ABI 8: emrt_mha_fwd reports through path_out (nullable) which kernel filled `probs` -- 0: the L x L probabilities (VALU kernel), 1: (row max,
 * 1 / row sum) in the first 2 L floats of each (batch, head) slab (MFMA kernel, bf16 / fp16) -- and emrt_mha_bwd takes that value as `path`: it runs the
 * matching backward or fails (it no longer re-derives the choice from its own operands). */
int emrt_mha_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, float* probs, int B, int M, int L, int D, float scale, float pdrop, const unsigned long long* seed, unsigned salt, int* path_out, int dtype, void* stream);
int emrt_mha_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* probs, const void* dout, int lddo, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int M, int L, int D, float scale, float pdrop, const unsigned long long* seed, unsigned salt, int path, int dtype, void* stream);

/* ---- bilinear F.interpolate (align_corners True/False), optional fused "+ add", strided (concat-slice) output,
 * optional fp32 NCHW output for the returned logits: paddle_EMRT.py:40,44,169,174,180,288-289,301; fcn_head.py:80 */
int emrt_resize_bilinear_fwd(const void* in, long long in_bs, int in_ld, int IH, int IW, void* out, long long out_bs, int out_ld, int OH, int OW, const void* add, long long add_bs, int add_ld, int N, int C, int align_corners, int out_nchw_f32, int dtype, void* stream);
/* ABI 6: out = resize([relu](BatchNorm_train(in))) in one pass over the RAW map of the producing conv -- emrt_bn_apply +
 * emrt_resize_bilinear_fwd without the normalised intermediate (nn.SyncBatchNorm -> ReLU -> F.interpolate, paddle_EMRT.py:164-175).
 * sums / count / eps / momentum / mean / invstd / run_* / gamma / beta / relu exactly as emrt_bn_apply takes them (sums complete, and
 * all-reduced by the caller for SyncBatchNorm).  Vector path only (C / 4 a divisor of 256, 16-byte aligned rows), an error otherwise.
 * Backward: emrt_resize_bilinear_bwd, then emrt_bn_bwd_reduce / emrt_bn_bwd_dx with the mask_* arguments. */
int emrt_bn_resize_bilinear_fwd(const void* in, long long in_bs, int in_ld, int IH, int IW, void* out, long long out_bs, int out_ld, int OH, int OW, int N, int C, int align_corners, const double* sums, double count, float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma, const float* beta, int relu, int dtype, void* stream);
size_t emrt_resize_bwd_workspace_bytes(int N, int C, int OH, int IW, int dout_nchw_f32);
int emrt_resize_bilinear_bwd(const void* dout, long long do_bs, int do_ld, int OH, int OW, void* din, long long di_bs, int di_ld, int IH, int IW, int N, int C, int align_corners, int dout_nchw_f32, void* workspace, int dtype, void* stream);
/* ABI 6: the decoder's pyramid token maps (k x k per scale, paddle_EMRT.py:281-291) resized to OH x OW, ALL scales in one launch per
 * direction.  tokens / dtokens: dense [N][sum k^2][C], the map of scale i starting at token sum_{j<i} k_j^2; outs[i] / douts[i]: NHWC maps with
 * row stride ld[i] and image stride bs[i] (channel slices of a concat buffer).  scales, outs, ld, bs: HOST arrays, nscales <= 4.  Vector path
 * only (C % 4 == 0, 16-byte aligned rows); backward additionally needs every map >= x4 smaller than the output per axis and writes every
 * token of dtokens (no zeroing needed).  Per scale these are emrt_resize_bilinear_fwd / _bwd launches of 8 ... 512 blocks. */
int emrt_pyramid_resize_fwd(const void* tokens, const int* scales, int nscales, void* const* outs, const int* out_ld, const long long* out_bs, int OH, int OW, int N, int C, int align_corners, int dtype, void* stream);
int emrt_pyramid_resize_bwd(const void* const* douts, const int* do_ld, const long long* do_bs, int OH, int OW, void* dtokens, const int* scales, int nscales, int N, int C, int align_corners, int dtype, void* stream);
/* nn.AdaptiveAvgPool2D(k), k in scales (host int[nscales], <= 4), all scales in one launch -> tokens [N][sum k^2][C]:
 * paddle_EMRT.py:62,70-78.  workspace (nullable; ABI 6: sized by emrt_adaptive_avgpool_workspace_bytes for the same H, W, N, C, scales, NOT
 * cleared by the caller): with it, maps whose largest bin has >= 512 pixels are pooled by several blocks per bin, each writing its partial
 * sum to its own slot, and a second launch adds a bin's parts in a fixed order -- no atomics, the same bits on every run -- instead of one
 * block per bin reading the whole map of the 1x1 scale through one CU.  The query returns 0 when the call would not use a workspace. */
size_t emrt_adaptive_avgpool_workspace_bytes(int H, int W, int N, int C, const int* scales, int nscales);
int emrt_adaptive_avgpool_fwd(const void* in, long long in_bs, int in_ld, int H, int W, void* out, long long out_bs, int out_ld, int N, int C, const int* scales, int nscales, float* workspace, size_t workspace_bytes, int dtype, void* stream);
int emrt_adaptive_avgpool_bwd(const void* dout, long long do_bs, int do_ld, void* din, long long di_bs, int di_ld, int H, int W, int N, int C, const int* scales, int nscales, int dtype, void* stream);
/* nn.MaxPool2D(3, 2, 1): paddle_vision_resnet.py:201; paddle_EMRT.py:84 (dense NHWC) */
int emrt_maxpool_fwd(const void* in, void* out, unsigned char* argmax, int N, int H, int W, int C, int k, int stride, int pad, int dtype, void* stream);
/* ABI 6: out = maxpool([relu](BatchNorm_train(in))) on the raw map (BatchNorm2D -> ReLU -> MaxPool2D: paddle_vision_resnet.py:199-201,
 * paddle_EMRT.py:84-91); BatchNorm arguments as emrt_bn_resize_bilinear_fwd.  C % 8 == 0, C <= 4096. */
int emrt_bn_maxpool_fwd(const void* in, void* out, unsigned char* argmax, int N, int H, int W, int C, int k, int stride, int pad, const double* sums, double count, float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma, const float* beta, int relu, int dtype, void* stream);
int emrt_maxpool_bwd(const unsigned char* argmax, const void* dout, void* din, int N, int H, int W, int C, int k, int stride, int pad, int dtype, void* stream);
/* model input: fp32 NCHW images (paddle_EMRT.py:252) -> NHWC compute dtype with c_out >= C channels (the extra ones zero: the
 * 3-channel image as an 8-channel map keeps the first convolutions on the GEMM kernels' 16-byte operand path) */
int emrt_nchw_to_nhwc(const float* in, void* out, int N, int C, int H, int W, int c_out, int dtype, void* stream);

/* ---- sliding-window inference glue, fp32 NCHW as the reference: src/api/infer.py:60-79 (crop / accumulate / count /
 * divide) and :150-155 (argmax; softmax dropped, it is monotonic).  origins_yx: HOST int[n][2], n <= 64 windows of
 * ch x cw inside the C x H x W image; final/count are caller-zeroed accumulators; uncovered pixels normalise to NaN. */
int emrt_crop_windows(const float* img, float* batch, const int* origins_yx, int n, int C, int H, int W, int ch, int cw, void* stream);
int emrt_window_accumulate(const float* logits, float* final, float* count, const int* origins_yx, int n, int C, int H, int W, int ch, int cw, void* stream);
int emrt_window_normalise(const float* final, const float* count, float* out, int C, int H, int W, void* stream);
int emrt_argmax_nchw(const float* logits, int* pred, int N, int C, int H, int W, void* stream);
/* multi-scale / flip test-time augmentation (infer.py:160-260): img[..., ::-1] and final += softmax(logit, axis=1) */
int emrt_flip_w(const float* in, float* out, long long rows, int W, void* stream);
int emrt_softmax_nchw_acc(const float* logits, float* acc, int N, int C, int H, int W, void* stream);

/* ---- loss: nn.CrossEntropyLoss(ignore_index, axis=1) on fp32 NCHW logits, int64 labels:
 * losses/mix_softmax_cross_entropy_loss.py:27-35.  result (device float[2]) = {mean loss, non-ignored count}. */
size_t emrt_ce_workspace_bytes(void);
int emrt_softmax_ce_fwd(const float* logits, const long long* labels, int N, int C, int H, int W, int ignore_index, float* result, void* workspace, void* stream);
int emrt_softmax_ce_bwd(const float* logits, const long long* labels, const float* result, const float* upstream, float weight, int N, int C, int H, int W, int ignore_index, float* dlogits, void* stream);
/* ABI 7: both heads of MixSoftmaxCrossEntropyLoss (losses/mix_softmax_cross_entropy_loss.py:29-35,44-51: CE(main) + AUX_WEIGHT * CE(aux) on the same
 * labels) at once -- one streaming launch over both logit tensors + one finalize that also forms total[0] = wa * loss_a + wb * loss_b; backward: one
 * launch for both gradients (dlogits_x = w_x * up_x * (softmax - onehot) / count; up_x NULL == 1).  Same per-pixel arithmetic as the single-head
 * entry points.  res_a / res_b: device float[2] = {mean loss, non-ignored count}. */
int emrt_softmax_ce_pair_fwd(const float* logits_a, const float* logits_b, const long long* labels, int N, int C, int H, int W, int ignore_index, float wa, float wb, float* res_a, float* res_b, float* total, void* workspace, void* stream);
int emrt_softmax_ce_pair_bwd(const float* logits_a, const float* logits_b, const long long* labels, const float* res_a, const float* up_a, const float* up_b, float wa, float wb, int N, int C, int H, int W, int ignore_index, float* dlogits_a, float* dlogits_b, void* stream);
int emrt_scalar_axpby(float* out, const float* a, float wa, const float* b, float wb, void* stream);

/* ---- optimizer: ClipGradByGlobalNorm + L2 decay + Momentum over one flat fp32 buffer, PolynomialDecay evaluated
 * on the device from a step counter: solver/optimizer.py:29-40, solver/lr_scheduler.py:244-248.
 * ranges: HOST int64 [nranges][2] element ranges whose lr is multiplied by range_mult (ParamAttr(learning_rate=0.1),
 * transformer_encoder_decoder.py:36-38,371-372).
 * mirror (nullable): bf16 / fp16 (mirror_dtype 1 / 2) copy of the parameters with the SAME indexing as `params`, refreshed by the
 * update itself: the forward GEMMs read their [OC][taps][C] operand straight from it, so the per-step weight re-pack only writes
 * the transposed dgrad copies (emrt_pack_weights with bwd_only = 1). */
size_t emrt_gradnorm_workspace_bytes(void);
int emrt_grad_clip_scale(const float* grads, long long n, float clip, float* state, void* workspace, void* stream);
int emrt_sgd_momentum_step(float* params, const float* grads, float* velocity, long long n, const float* clip_state, const long long* step, float base_lr, float end_lr, float power, long long decay_steps, float momentum, float weight_decay, const long long* ranges, int nranges, float range_mult, float* lr_out, void* mirror, int mirror_dtype, void* stream);
/* ---- evaluation metric counts (src/utils/metrics.py:20-59 calculate_area; val.py:197-209): out[3][num_classes] (int64, ACCUMULATED into: the caller zeroes it
 * once per evaluation) += per-class pixel counts of (pred == label), pred, label over the pixels whose label != ignore_index; predictions / labels outside
 * [0, num_classes) are counted nowhere.  pred int32 (the argmax map of emrt_argmax / the sliding-window engine), label int64 (label_is_int64) or int32. */
int emrt_segmentation_areas(const int* pred, const void* label, int label_is_int64, long long n, int num_classes, int ignore_index, long long* out, void* stream);
int emrt_counter_add(long long* counter, long long delta, void* stream);
/* fp32 master weights -> compute-dtype forward copy [OC][taps][C] and transposed dgrad copy [C][taps][OC];
 * desc_dev: DEVICE int64 [ndesc][8] = {src_off, fwd_off|-1, bwd_off|-1, OC, taps, C, first 32x32 tile, first 64x64 tile};
 * total_tiles / total_tiles64 = sum over descriptors of taps * ceil(OC/32) * ceil(C/32) (resp. /64).  bwd_only = 1 skips the forward
 * copies (the optimizer keeps them current through its `mirror`) and transposes 64x64 tiles FROM them. */
int emrt_pack_weights(const float* master, void* packed, const long long* desc_dev, int ndesc, long long total_tiles, long long total_tiles64, int bwd_only, int dtype, void* stream);

/* ---- small streaming ops: with_pos_embed adds, nn.Dropout/nn.Dropout2D (mode 1), ReLU/dropout backward masks,
 * F.sigmoid, casts: transformer_encoder_decoder.py:115-122,154-161,250-263,273-280,466; paddle_EMRT.py:208; fcn_head.py:65 */
int emrt_add(const void* a, const void* b, void* out, long long n, long long period, int dtype, void* stream);
int emrt_add3d(const void* a, long long a_bs, long long a_rs, const void* b, long long b_bs, long long b_rs, void* out, long long out_bs, long long out_rs, long long B, long long rows, long long cols, int dtype, void* stream);
/* [B][n_i][C] dense parts -> dense whole [B][sum n_i][C] (split = 0) or back (split = 1), one launch; parts / n: HOST arrays (<= 8):
 * the pyramid-pooling token concat of paddle_EMRT.py:70-78 and its backward */
int emrt_concat_tokens(void* const* parts, const int* n, int nparts, void* whole, int B, int C, int split, int dtype, void* stream);
int emrt_acc3d(void* dst, long long dst_bs, long long dst_rs, const void* src, long long src_bs, long long src_rs, long long B, long long rows, long long cols, int dtype, void* stream);
int emrt_add_f32row(const void* a, const float* row, void* out, long long n, long long period, int dtype, void* stream);
/* out[r][:] = a[r][:] + rows[l][:] for the token rows r of level l (level_start: HOST array of L <= 4 increasing first rows, level_start[0] == 0; rows fp32 [L][C]):
 * pos = sine + level_embed[l] over all levels of the pyramid in one launch (transformer_encoder_decoder.py:447-448) */
int emrt_add_f32row_levels(const void* a, const float* rows, void* out, const int* level_start, int L, int Lv, int C, int dtype, void* stream);
int emrt_dropout_fwd(const void* x, void* y, long long n, float p, const unsigned long long* seed, unsigned salt, int mode, long long hw, int C, int dtype, void* stream);
/* dx = dy * dropmask(p, seed, salt) / (1 - p) * (relu_out > 0); either mask optional (p == 0 / relu_out == NULL).  ABI 7: p > 0 with seed == NULL and
 * relu_out given: relu_out is the stored output of emrt_conv2d_drop, whose sign carries BOTH masks: dx = relu_out > 0 ? dy / (1 - p) : 0. */
int emrt_mask_bwd(const void* dy, const void* relu_out, void* dx, long long n, float p, const unsigned long long* seed, unsigned salt, int mode, long long hw, int C, int dtype, void* stream);
int emrt_sigmoid_fwd(const float* x, float* y, long long n, void* stream);
int emrt_sigmoid_bwd(const float* y, const float* dy, float* dx, long long n, void* stream);
int emrt_cast(const void* in, void* out, long long n, int direction, int dtype, void* stream);
int emrt_memset(void* ptr, int value, size_t bytes, void* stream);
/* device -> device copy on the caller's stream: staging a batch into the buffers a captured hipGraph reads (train.py:141-146's `images, labels` of a step) */
int emrt_memcpy(void* dst, const void* src, size_t bytes, void* stream);

/* ---- HIP events on the caller's stream (bench.py roofline timing) ---- */
int emrt_event_create(void** ev);
int emrt_event_record(void* ev, void* stream);
int emrt_event_elapsed_ms(void* start, void* stop, float* ms);
int emrt_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* EMRT_HIP_H */
