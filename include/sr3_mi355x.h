/* libsr3_mi355x.so -- C ABI of the MI355X-native SR3 / DDPM iterative-refinement engine.
 *
 * The reference (Janspiry/Image-Super-Resolution-via-Iterative-Refinement) is pure PyTorch and has
 * no FFI of its own; the drop-in boundary is its Python surface (model.networks.define_G,
 * GaussianDiffusion, DDPM -- see INTEGRATION.md).  This header is the C layer underneath that
 * surface: plain pointers and sizes only, no torch types.  Each entry cites the reference code it
 * replaces (paths relative to the reference root).
 *
 * Conventions
 *  - every pointer named *_dev / every `const float*` tensor argument is a DEVICE pointer
 *    (tensor.data_ptr()); `stream` is a hipStream_t passed as void* (0 = default stream);
 *  - all calls are asynchronous on `stream`, never synchronise, never allocate or free device
 *    memory (workspaces are sized by a query and passed in), and are hipGraph-capturable;
 *  - return value: 0 ok, >0 a hipError_t, <0 an engine error (SR3_E_*); sr3_last_error() returns a
 *    thread-local message; no C++ exception crosses the ABI;
 *  - public tensors (x, cond, eps, z) are NCHW fp32 as in the reference; internal activations are
 *    NHWC fp32 inside the workspace.
 */
#ifndef SR3_MI355X_H
#define SR3_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR3_ABI_VERSION 1

#define SR3_E_BADARG (-1)
#define SR3_E_UNSUPPORTED (-2)
#define SR3_E_ALIGN (-3)
#define SR3_E_NOMEM (-4)
#define SR3_E_STATE (-5)

#define SR3_VARIANT_SR3 0  /* model/sr3_modules: continuous noise-level conditioning */
#define SR3_VARIANT_DDPM 1 /* model/ddpm_modules: integer-timestep conditioning      */

/* Mirrors the arguments of UNet.__init__ (model/sr3_modules/unet.py:162-173,
 * model/ddpm_modules/unet.py:148-159) as filled by define_G (model/networks.py:91-101). */
typedef struct sr3_unet_desc {
  int variant;
  int in_channel, out_channel, inner_channel, norm_groups;
  int n_mults;
  int channel_mults[8];
  int n_attn_res;
  int attn_res[8];
  int res_blocks;
  int image_size;
} sr3_unet_desc;

/* One entry of the parameter table: where a reference state_dict tensor lives in the packed
 * parameter arena.  pack: 0 = copied as is (row-major), 1 = conv weight OIHW -> OHWI. */
typedef struct sr3_param_info {
  char name[128]; /* reference key without the "denoise_fn." prefix, e.g. downs.1.res_block.block1.block.3.weight */
  int ndim;
  int shape[4];   /* reference shape (OIHW for convs) */
  int pack;
  size_t offset;  /* float offset inside the arena */
  size_t numel;
} sr3_param_info;

typedef struct sr3_plan sr3_plan;

int sr3_version(void);
const char* sr3_last_error(void);
/* Device self-test of the 3 x bf16 operand split every SPLIT kernel rests on (x == h + m + l exactly, 2^21 fp32 patterns):
 * scratch_dev = one int of device memory, *mismatches (host) = offending elements; SR3_E_UNSUPPORTED when it is not 0.
 * (No reference counterpart: the reference computes in IEEE fp32; this guards the claim that the split path is fp32-class.) */
int sr3_selftest_split3(int* scratch_dev, int* mismatches, void* stream);

/* UNet construction (model/sr3_modules/unet.py:162-233): builds the layer list, the parameter
 * table and the activation plan.  No device work. */
int sr3_plan_create(const sr3_unet_desc* desc, sr3_plan** out);
void sr3_plan_destroy(sr3_plan* plan);
int sr3_plan_num_params(const sr3_plan* plan);
int sr3_plan_param_info(const sr3_plan* plan, int index, sr3_param_info* out);
size_t sr3_plan_param_floats(const sr3_plan* plan);
/* ordered list of kernels one forward launches, for inspection / DESIGN.md: returns count */
int sr3_plan_num_ops(sr3_plan* plan, int batch);
/* one entry of that list: kind as in sr3_unet_forward_profile's op_kind / 10 * 10 (10 embed, 20 input conv, 30
 * statistics, 40 GroupNorm fold, 50 convolution, 60 attention, 70 output block); for convolutions the tile
 * configuration the plan picked (1-4 im2col kernel, 5-10 halo-tile kernel, see sr3_conv_f32), the split-K factor, the
 * geometry and what is fused into the launch.  Host-only (no device work): lets tools and tests inspect the plan. */
typedef struct sr3_op_info {
  int kind, tile_cfg, ksplit;
  int ksize, stride, upsample;
  int cin, cout, h_out, w_out;          /* attention: cin = cout = channels, h_out = tokens */
  int fused_res_conv_cin;               /* > 0: the 1x1 res_conv of that many input channels runs inside this launch */
  int fused_output_stats;               /* 1: the launch also writes the next GroupNorm's partial statistics */
  double flops;
} sr3_op_info;
int sr3_plan_op_info(sr3_plan* plan, int batch, int index, sr3_op_info* out);
/* plan option fork_side: *side_id >= 0 -- op `index` is launched on the plan's side stream, forked from the caller's stream by an event
 * where the op sits in the list; *wait_id >= 0 -- the caller's stream waits for the side op of that id before this op (its consumer).
 * Both -1 for every op of a plan without the option.  Host-only. */
int sr3_plan_op_side(sr3_plan* plan, int batch, int index, int* side_id, int* wait_id);
/* algorithmic FLOPs (contractions only) of one forward for `batch` images */
double sr3_plan_forward_flops(sr3_plan* plan, int batch);
/* tuning knobs: key in {"fuse_stats", "fuse_res", "tile_cfg", "ksplit", "keep_all", "split_bf16", "winograd",
 * "wino_split", "wino_split8", "wino2", "gemm_split", "gemm2", "gemm_s2", "gemm_n64", "fork_side", "gemm_wpre", "gemm_tile", "fold_fuse", "wgrad_split", "attn_split",
 * "loss_l2"};
 * returns previous value.
 * wino_split (default 1): the Winograd convolutions that run on the kernel's one-image tile (maps >= 16x16) use its 3 x bf16
 *   split instantiation: every fp32 operand as x = h + m + l (three bf16 terms, each residual exact in fp32), every product as
 *   the six bf16 MFMA products hh, hm, mh, mm, hl, lh with fp32 accumulation (v_mfma_f32_32x32x16_bf16) -- fp32-class results
 *   (dropped terms <= 2^-24 of a product; measured not less accurate than the fp32-MFMA instantiation on every layer shape,
 *   tests/test_gpu_ops.py), not the bit pattern of the fp32 MFMA.  The derived buffer then holds both forms of every filter
 *   (sr3_plan_derived_bytes grows 2.5x; re-bind after toggling).  0: v_mfma_f32_32x32x2_f32 everywhere.
 * gemm_split (default 1): the same 3 x bf16 split arithmetic for the convolutions of the im2col kernel (every 1x1 conv --
 *   res_conv, attention qkv / out -- and the stride-2 Downsample convs), operands split while they are staged into LDS.
 *   0: v_mfma_f32_32x32x2_f32.
 * gemm_tile (default 0 = automatic; A/B runs): force im2col tile 1-4 (128x128 / 128x64 / 64x64 / 64x128) on every convolution of that
 *   kernel -- how profiles/r04f_gemm_split_sweep.txt timed the tiles inside the forward.
 * wino_split8 (default 1, round 5): the four-image tile of the 8x8 maps on its 3 x bf16 split instantiation too (no dropout form:
 *   a training forward's dropout convs on 8x8 maps keep the fp32 MFMA).
 * wino2 (default 1, round 6): the wino_split convolutions of the one-image tile without dropout -- every 3x3 stride-1 convolution on
 *   maps >= 16 wide in an inference plan, block1 / Upsample convs and the data gradients in a training plan -- run as TWO independent
 *   four-wave workgroups per CU on an 8 x 16 pixel tile (conv3x3_wino2.hip; reported as tile 13): same arithmetic and derived filters,
 *   a wave owns one transform column and all four rows.  0: the 8-wave kernel of conv3x3_wino.hip everywhere (tile 12).
 * gemm2 (default 1, round 6; needs gemm_split): 1x1 stride-1 convolutions with Cout % 128 == 0, channel counts % 32 == 0 and
 *   B * H * W % 64 == 0 (res_conv, attention qkv / out of the BASELINE networks) run as a plain GEMM on the same 3 x bf16 split
 *   arithmetic (gemm1x1.hip; reported as tile 22): 64 x 128 tile, weights pre-split in MFMA fragment order in the derived buffer
 *   (6 bytes per weight: sr3_plan_derived_bytes grows; re-bind after toggling) and read straight from global memory, the A rows
 *   split once per 128 output channels; a training plan's 1x1 data gradients use it too.  0: the im2col kernel (tiles 14-17).
 * gemm_s2 (default 1, round 6; needs gemm2): Downsample's 3x3 stride-2 pad-1 convolutions with Cout % 128 == 0 (one source, no
 *   activation, even maps) run on the same kernel's stride-2 form (also reported as tile 22): the GEMM's k-steps walk (32-channel
 *   chunk, tap), the A row of a tap is the NHWC row of the shifted input pixel, padding rows are zeroed where they are staged;
 *   54 bytes per weight in the derived buffer.  0: the im2col kernel (tile 16).
 * gemm_n64 (default 1, round 6; needs gemm2): the layers with Cout % 128 != 0 and Cout % 64 == 0 (res_conv of the first resolution level,
 *   Downsample 64 -> 64 with gemm_s2, data gradients towards 64 / 192 channels) on the same kernel's 64 x 64 tile, its four waves 2 x 2.
 *   0: they keep the im2col kernel (tile 16; Downsample 64 -> 64 its fp32 form, tile 2).
 * fork_side (default 0; A/B knob, inference plans): every unsplit res_conv is emitted in front of its block's first conv and launched
 *   on a side stream the plan owns (fork / join by events: a parallel branch once the forward is captured into a graph), block2's conv
 *   waits for it; the embedding MLP runs beside the input conv the same way.  Results are bit-identical (same kernels, same operands).
 * fold_fuse (default 1, round 6): the GroupNorm fold of a consumer is done by the kernel that completes its last source where that
 *   is a split-K reduce or a stand-alone statistics pass (one workgroup per (image, consumer group), no atomics): those fold
 *   launches leave the launch list (sr3_plan_num_ops shrinks), the conv outputs are bit-identical, the folded (scale, shift) pairs
 *   differ in the order of their double-precision sums only.  0: every fold is a launch of its own.
 * gemm_wpre (default 0): the im2col split tiles read their weights pre-split AND in MFMA fragment order straight from the derived
 *   buffer (tiles 18-21; round 6's form, no LDS staging of the weights) instead of splitting them while staging (14-17, what a plan
 *   runs); measured slower in every layout tried (profiles/r05c_*, profiles/r06_gemm_wpre_fragment_major.txt), kept as an A/B knob.
 * wgrad_split (default 1, round 5; training): weight gradients of the layers with more than 64 input AND output channels on
 *   v_mfma_f32_32x32x16_bf16 with 3-way split operands (wgrad.hip), gated by the batch-64 gradient tests against float64 autograd;
 *   0: the fp32-MFMA weight-gradient kernels.  No rebuild of the plan.
 * attn_split (default 1, round 5): SelfAttention's two contractions (Q K^T, P V) on the 3 x bf16 split instantiation of the
 *   staging-free kernel, gated against float64 like the convolutions; 0: v_mfma_f32_32x32x2_f32.  No rebuild of the plan.
 * loss_l2 (default 0): sr3_train_step uses nn.MSELoss(reduction='sum') instead of nn.L1Loss(reduction='sum')
 *   (GaussianDiffusion(loss_type='l2'), model/sr3_modules/diffusion.py:84-90).
 * split_bf16 (default 0, experimental; needs -DSR3_EXPERIMENTS, refused otherwise): run the halo-tile 3x3 convolutions of the inference plan on
 *   v_mfma_f32_32x32x16_bf16 with every fp32 operand split into three bf16 terms (x = h + m + l) and the six
 *   products hh, hm, mh, mm, hl, lh accumulated in fp32 -- fp32-class accuracy (dropped terms <= 2^-23 of a
 *   product), not the bit pattern of the fp32 MFMA.  Operands are split while they are staged into LDS (a pre-split
 *   weight copy was measured slower: 1.5x the L2->LDS bytes). */
int sr3_plan_set_option(sr3_plan* plan, const char* key, int value);

/* Debug taps: where each top-level layer output (downs.i / mid.i / ups.i, NHWC) lives inside the
 * workspace after a forward.  Only meaningful with option keep_all = 1 (no buffer reuse). */
int sr3_plan_num_taps(sr3_plan* plan);
int sr3_plan_tap_info(sr3_plan* plan, int index, char* name, int name_len, size_t* offset, int* C, int* H, int* W);

/* Workspace bytes for sr3_unet_forward at this batch size (activations, statistics, FiLM table,
 * split-K slabs). */
/* Derived weights.  The inference plan runs its 3x3 stride-1 convolutions as Winograd F(2x2,3x3) (plan option
 * "winograd", default 1), which reads the transformed filters U = G g G^T from a caller-owned device buffer of
 * sr3_plan_derived_bytes bytes (16/9 of the 3x3 weights).  Bind it once (the pointer is kept, and baked into captured
 * graphs), and re-run sr3_plan_prepare_derived on the stream whenever the parameter arena changed (checkpoint load,
 * optimizer step): ~55 small launches.
 * STALE FILTERS FAIL LOUDLY: sr3_unet_forward / sr3_train_step return SR3_E_BADARG when the plan needs the filters and
 * (a) none is bound, (b) the buffer was never prepared, (c) it was prepared from a different `params` pointer than the
 * call's, (d) one of the plan options winograd / tile_cfg / split_bf16 / wino_split was set since (wino_split also changes
 * sr3_plan_derived_bytes and un-binds a buffer that is now too small), or (e) sr3_plan_invalidate_derived was called
 * after the last prepare.  The library cannot see writes to the arena (sr3_adam_step takes no plan): a caller that
 * updates parameters in place calls sr3_plan_invalidate_derived right after the update and prepares again before the
 * next forward. */
size_t sr3_plan_derived_bytes(const sr3_plan* plan);
int sr3_plan_bind_derived(sr3_plan* plan, void* buffer, size_t bytes);
int sr3_plan_prepare_derived(sr3_plan* plan, const float* params, void* stream);
int sr3_plan_invalidate_derived(sr3_plan* plan);
size_t sr3_workspace_bytes(sr3_plan* plan, int batch);

/* UNet.forward (model/sr3_modules/unet.py:235-259, model/ddpm_modules/unet.py:220-243).
 *   x_nchw    : (B, in_channel - cond_channels, S, S)  the noisy image
 *   cond_nchw : (B, cond_channels, S, S) or NULL -- the conditioning image; the engine consumes the
 *               pair as the virtual concat torch.cat([cond, x], 1) (model/sr3_modules/diffusion.py:157)
 *   noise_level : (B) fp32, SR3 variant (the (B,1) tensor of diffusion.py:153-154), else NULL
 *   timestep  : (B) int64, DDPM variant, else NULL
 *   freq      : (inner_channel/2) fp32 frequency table (PositionalEncoding / TimeEmbedding.inv_freq)
 *   level_table/step_dev : optional graph-replay source of the conditioning value: when step_dev is
 *               non-NULL the level is level_table[*step_dev + 1] (SR3) or the timestep *step_dev (DDPM)
 *   params    : packed parameter arena (see sr3_plan_param_info)
 *   eps_out_nchw : (B, out_channel, S, S) */
int sr3_unet_forward(sr3_plan* plan, const float* x_nchw, const float* cond_nchw, int cond_channels,
                     const float* noise_level, const int64_t* timestep, const float* freq,
                     const float* level_table, const int* step_dev, const float* params,
                     void* workspace, size_t workspace_bytes, float* eps_out_nchw, int batch,
                     void* stream);

/* Same forward, run eagerly with a hipEvent pair around every launch of the plan (events are
 * recorded on `stream`).  Measurement aid for bench.py's roofline leg; not capturable.
 * One entry per kernel launch.  op_kind: 10 embed+FiLM, 20 input conv, 30 GN statistics, 40 GN fold,
 * 51..54 im2col implicit-GEMM conv (tile config), 55/56 halo-tile 3x3 conv k_conv3x3_halo<2,2,false> /
 * <4,1,false>, 57/58 the same with the fused 1x1 res_conv segment (<.,.,true>), 59 split-K reduce,
 * 60 attention, 70 output Block.  op_flops: algorithmic FLOPs of contractions (0 for HBM-bound helpers). */
int sr3_unet_forward_profile(sr3_plan* plan, const float* x_nchw, const float* cond_nchw, int cond_channels,
                             const float* noise_level, const int64_t* timestep, const float* freq,
                             const float* params, void* workspace, size_t workspace_bytes, float* eps_out_nchw,
                             int batch, void* stream, int max_ops, float* op_ms, int* op_kind, double* op_flops,
                             int* n_ops);

/* Fused elementwise tail of p_mean_variance + p_sample (model/sr3_modules/diffusion.py:141-149,
 * 162-174; model/ddpm_modules/diffusion.py:151-198), in place on x:
 *   x0 = a[t] x - b[t] eps ; clamp(-1,1) ; mean = c1[t] x0 + c2[t] x ; x = mean + sigma[t] z
 * tables (length T, fp32): a = sqrt_recip_alphas_cumprod, b = sqrt_recipm1_alphas_cumprod,
 * c1/c2 = posterior_mean_coef1/2, sigma[t] = exp(0.5 posterior_log_variance_clipped[t]), sigma[0] = 0.
 * t is *step_dev if non-NULL, else t_per_sample[b] if non-NULL, else step_host.  z may be NULL (= 0). */
int sr3_p_sample_step(float* x_nchw, const float* eps_nchw, const float* z_nchw, const float* tab_a,
                      const float* tab_b, const float* tab_c1, const float* tab_c2, const float* tab_sigma,
                      const int* step_dev, const int64_t* t_per_sample, int step_host, int batch,
                      int elems_per_image, void* stream);
/* The same update with the reference's `clip_denoised` switch (p_mean_variance, model/sr3_modules/diffusion.py:162-163,
 * model/ddpm_modules/diffusion.py:184-185): clip_denoised == 0 skips the clamp of x0; != 0 is sr3_p_sample_step. */
int sr3_p_sample_step_ex(float* x_nchw, const float* eps_nchw, const float* z_nchw, const float* tab_a,
                         const float* tab_b, const float* tab_c1, const float* tab_c2, const float* tab_sigma,
                         const int* step_dev, const int64_t* t_per_sample, int step_host, int batch,
                         int elems_per_image, int clip_denoised, void* stream);
/* *step_dev -= 1 on the stream (loop counter of p_sample_loop, diffusion.py:193, for graph replay) */
int sr3_step_decrement(int* step_dev, void* stream);

/* One WHOLE iteration of the reference's reverse loop (model/sr3_modules/diffusion.py:190-196 `for i in reversed(range(T)): img = p_sample(img, i, ...)`
 * with p_sample = :169-174, p_mean_variance :151-167; model/ddpm_modules/diffusion.py:200-215) as one capturable call:
 *   eps = UNet(cat(cond, x), level(t))  ;  x <- p_sample update of (x, eps, z, t)  ;  t <- t - 1
 * = sr3_unet_forward + sr3_p_sample_step_ex + sr3_step_decrement, with the last two inside the output convolution's kernel (the thread
 * that produces an element of eps updates the same element of x; separately rounded operations, bit-identical to the three-call form):
 * two kernel nodes fewer per replayed step.
 *   x_nchw      : [B, C, S, S] the image, in / out
 *   step2_dev   : TWO ints.  step2_dev[1] = t of this step on entry (the caller sets it to T - 1 before the first step) and t - 1 on
 *                 completion; step2_dev[0] is scratch (the step's first kernel copies t there for its last one).  t must stay >= 0.
 *   level_table : SR3: level = level_table[t + 1] (sqrt_alphas_cumprod_prev); DDPM: ignored (the timestep is t)
 *   z_nchw      : the step's noise or NULL (= 0); tab_*: the schedule tables of sr3_p_sample_step (sigma[0] = 0 replaces `t > 0`)
 *   eps_out_nchw: NULL, or where to also store eps (parity checks) */
int sr3_reverse_step(sr3_plan* plan, float* x_nchw, const float* cond_nchw, int cond_channels, const float* freq,
                     const float* level_table, int* step2_dev, const float* params, void* workspace, size_t workspace_bytes,
                     const float* z_nchw, const float* tab_a, const float* tab_b, const float* tab_c1, const float* tab_c2,
                     const float* tab_sigma, int clip_denoised, float* eps_out_nchw, int batch, void* stream);

/* q_sample (model/sr3_modules/diffusion.py:212-219; model/ddpm_modules/diffusion.py:259-267):
 * out = ca[b] * x0 + cb[b] * z */
int sr3_q_sample(const float* x0, const float* z, const float* ca, const float* cb, int batch,
                 int elems_per_image, float* out, void* stream);

/* ---- training step ------------------------------------------------------------------------- */

/* Workspace of sr3_train_step (activations kept for the backward, their gradient mirror, scratch). */
size_t sr3_train_workspace_bytes(sr3_plan* plan, int batch, int cond_channels);

/* One training step up to the gradients: `l_pix = netG(data); l_pix.backward()` of
 * DDPM.optimize_parameters (model/model.py:50-54) over GaussianDiffusion.p_losses
 * (model/sr3_modules/diffusion.py:221-246, model/ddpm_modules/diffusion.py:278-294):
 *   x_noisy = q_ca[b] * hr + q_cb[b] * z ; eps = UNet(cat(cond, x_noisy), level) ;
 *   *loss_sum_out = sum |z - eps| ; grads = d(grad_scale * loss_sum) / d params
 * (grad_scale = 1 / (b*c*h*w), model.py:52-53).  The random draws (t / gamma, z) are made by the caller
 * (torch / numpy RNG, as in the reference) and passed in.  `grads` has the layout of the parameter
 * arena and is overwritten.  dropout_p > 0 applies nn.Dropout(p) between Swish and the conv of every
 * block2 (unet.py:83-88,100-101) with a counter-based mask: activated element i (NHWC linear index) of
 * the block with FiLM row offset k is kept iff hash32(i*0x9E3779B9 + dropout_seed + (k+1)*0x632BE5AB) >=
 * p*2^32 and scaled by 1/(1-p); the mask is regenerated, never stored, by the backward.
 * Gradient-ready marks (data-parallel overlap): mark_offsets[k] (descending arena offsets) / mark_events[k]
 * (hipEvent_t): event k is recorded on `stream` as soon as the gradient of every parameter at arena offset
 * >= mark_offsets[k] has been enqueued, so a communication stream can all-reduce that tail bucket while the
 * rest of the backward still runs (replaces nn.DataParallel's reduce_add, model/networks.py:113-115). */
int sr3_train_step(sr3_plan* plan, const float* hr_nchw, const float* cond_nchw, int cond_channels,
                   const float* z_nchw, const float* q_ca, const float* q_cb, const float* noise_level,
                   const int64_t* timestep, const float* freq, const float* params, float* grads,
                   void* workspace, size_t workspace_bytes, float* loss_sum_out, float grad_scale, float dropout_p,
                   unsigned dropout_seed, int n_marks, const size_t* mark_offsets, void* const* mark_events,
                   int batch, void* stream);

/* torch.optim.Adam step (model/model.py:39-40,55; defaults beta 0.9/0.999, eps 1e-8, no weight decay)
 * fused over the whole arena; `step` is the 1-based step count for the bias corrections. */
int sr3_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                  float beta1, float beta2, float eps, int step, void* stream);

/* ---- per-op entry points (unit tests, micro-benchmarks) ------------------------------------ */

/* Block / Conv2d / Downsample / Upsample / res_conv / qkv / out as one implicit-GEMM call:
 * NHWC in/out, input = virtual concat (src0|src1), optional x2 nearest upsample, stride 1|2,
 * ksize 1|3 (pad ksize/2), prologue act 0 none | 1 x*scale+shift | 2 silu(x*scale+shift) with
 * ss[B][Cin][2]; epilogue + bias + film[b*film_stride+n] + residual (res0|res1 concat view).
 * weights OHWI.  tile_cfg/ksplit 0 = auto (direct kernels only); tile_cfg 11 = Winograd F(2x2,3x3) (3x3 stride 1, H and W
 * multiples of 16, or 8x8 maps with B % 4 == 0 -- four images per workgroup tile, split-K only; the transformed filters are
 * derived into `scratch` by this entry point); tile_cfg 12 = the same kernel's 3 x bf16 split instantiation (one-image tile
 * only, and the four-image tile of the 8x8 maps; what plan option wino_split selects there), 13 = the same arithmetic as two
 * four-wave workgroups per CU on an 8 x 16 pixel tile (conv3x3_wino2.hip; W >= 16 and a multiple of 16, H a multiple of 8, no
 * dropout form; what plan option wino2 -- default on -- selects on maps >= 16 wide); tile_cfg 1-4 = the
 * im2col kernel's 128x128 / 128x64 / 64x64 / 64x128 tiles on the exact-fp32 MFMA, 14-17 = the same tiles on the 3 x bf16 split
 * instantiation with both operands split while they are staged (what plan option gemm_split selects: what a plan runs), 18-21 =
 * the same with the weights pre-split into bf16 planes in MFMA fragment order and read straight from global memory (plan option
 * gemm_wpre, default off: a plan keeps the planes in its derived buffer; this entry point derives them into `scratch`; results
 * are bit-identical to 14-17); 22 = the plain GEMM kernel of gemm1x1.hip (what plan options gemm2 / gemm_s2 -- default on -- select: 1x1
 * stride 1, or 3x3 stride 2 with one source, act 0 and an even map; no upsampling, Cout % 64 == 0, C0 and C1 % 32 == 0,
 * B * Ho * Wo % 64 == 0, Ho * Wo % 32 == 0, act 0 | 1; anything else is refused with "does not fit"; same pre-split weights as
 * 18-21, derived into `scratch`).
 * scratch: split-K slabs (+ the Winograd filters for tile_cfg 11-13, the pre-split weights for 18-22), sized by
 * sr3_conv_scratch_bytes. */
int sr3_conv_f32(const float* src0, int C0, const float* src1, int C1, int B, int Hs, int Ws, int ups,
                 int stride, int ksize, int Cout, const float* w_ohwi, const float* bias, const float* ss,
                 int act, const float* film, int film_stride, const float* res0, int RC0, const float* res1,
                 int RC1, float* out, double* out_stats, int tile_cfg, int ksplit, void* scratch,
                 size_t scratch_bytes, void* stream);
/* ResnetBlock tail in one launch (unet.py:105-110): out = conv3x3(act(src0|src1)) + bias + film
 *   + conv1x1(x2_src0|x2_src1; x2_w [Cout][x2_C0+x2_C1]) + x2_bias  -- block2's conv with `res_conv(x)`
 * accumulated as a second K-segment of the same output tile (halo kernel: tile_cfg 0 | 5 | 6). */
int sr3_block_conv_f32(const float* src0, int C0, const float* src1, int C1, int B, int H, int W, int Cout,
                       const float* w_ohwi, const float* bias, const float* ss, int act, const float* film,
                       int film_stride, const float* x2_src0, int x2_C0, const float* x2_src1, int x2_C1,
                       const float* x2_w, const float* x2_bias, float* out, double* out_stats, int tile_cfg,
                       int ksplit, void* scratch, size_t scratch_bytes, void* stream);
/* Train-mode `Block` (unet.py:80-91 with nn.Dropout active, :86): out = conv3x3(dropout(act(src0))) + bias + film
 *   [+ residual res0] [+ conv1x1(x2_src0|x2_src1) + x2_bias], the op sr3_train_step launches for every block2.
 * Mask: NHWC element i of the activated input is kept iff hash32(i * 0x9E3779B9 + drop_seed) >= drop_p * 2^32 and
 * scaled by 1 / (1 - drop_p) (drop_seed is the per-layer seed).  Single source, no upsampling; x2_* may be NULL.
 * tile_cfg 11 / 12 run the Winograd kernel's dropout instantiations (fp32 MFMA / 3 x bf16 split; what sr3_train_step uses on
 * maps >= 16x16 with wino_split = 0 / 1; no x2 segment:
 * scratch then also holds the transformed filters, as for sr3_conv_f32 -- sr3_conv_scratch_bytes accounts for them). */
int sr3_conv_dropout_f32(const float* src0, int C0, int B, int H, int W, int Cout, const float* w_ohwi,
                         const float* bias, const float* ss, int act, const float* film, int film_stride,
                         const float* res0, int RC0, const float* x2_src0, int x2_C0, const float* x2_src1, int x2_C1,
                         const float* x2_w, const float* x2_bias, float* out, double* out_stats, int tile_cfg,
                         int ksplit, void* scratch, size_t scratch_bytes, unsigned drop_seed, float drop_p,
                         void* stream);
/* The keep threshold and scale every dropout kernel derives from p: floor(p * 2^32) and 1 / (1 - p), evaluated in double
 * on the decimal value the fp32 argument stands for (0.2f -> 0.2), as nn.Dropout's p is a Python float.  Host only. */
unsigned sr3_dropout_threshold(float drop_p, float* scale_out);
size_t sr3_conv_scratch_bytes(int B, int Ho, int Wo, int Cin, int Cout, int ksize, int tile_cfg, int ksplit);
/* nn.GroupNorm statistics (unet.py:84,119) as PARTIAL per-(image, channel) {sum, sumsq} in double of
 * an NHWC tensor: stat[B][T][C][2] with T = sr3_groupnorm_stats_slices(B, HW, C).  Plain stores (no
 * atomics, no zeroing needed); summed in a fixed order by the fold => bitwise reproducible. */
int sr3_groupnorm_stats_slices(int B, int HW, int C);
int sr3_groupnorm_stats_f32(const float* x_nhwc, int B, int HW, int C, double* stat, void* stream);
/* T of the partial statistics sr3_conv_f32 writes into out_stats (0: this geometry cannot fuse them) */
int sr3_conv_stats_slices(int B, int Hs, int Ws, int ups, int Cin, int Cout, int tile_cfg, int ksplit);
/* fold partial statistics of the concat (stat0|stat1) with gamma/beta into ss[B][C0+C1][2] */
int sr3_groupnorm_fold_f32(const double* stat0, int C0, int T0, const double* stat1, int C1, int T1, int B, int HW,
                           int groups, const float* gamma, const float* beta, float eps, float* ss, void* stream);
/* SelfAttention core (unet.py:127-139): qkv NHWC [B][N][3C] -> out [B][N][C] */
int sr3_attention_f32(const float* qkv, int B, int N, int C, float* out, void* stream);
/* ... with split != 0: QK^T and PV as six bf16 MFMA products of 3-way split fp32 operands, fp32 accumulation (plan option
 * attn_split; fp32-class results, gated against float64 in tests/), where the shape takes the staging-free kernel */
int sr3_attention_ex_f32(const float* qkv, int B, int N, int C, float* out, int split, void* stream);
/* backward of the attention core (autograd of unet.py:127-139): dqkv [B][N][3C] from qkv, d(out) [B][N][C]; out_fwd
 * (the forward output) may be NULL when N <= ~480 -- larger N use a key-blocked pass that reads it */
int sr3_attention_bwd_f32(const float* qkv, const float* dout, const float* out_fwd, int B, int N, int C, float* dqkv,
                          void* stream);
/* ... bitwise reproducible: every 32-query block hands its dK / dV contribution to a slab of its own in `scratch`
 * (sr3_attention_bwd_scratch_bytes(B, N, C) = ceil(N / 32) * B * N * 2C floats) and a second kernel sums the slabs in block
 * order -- no atomics, no memset (sr3_attention_bwd_f32 adds with fp32 atomics).  What sr3_train_step runs. */
size_t sr3_attention_bwd_scratch_bytes(int B, int N, int C);
int sr3_attention_bwd_ex_f32(const float* qkv, const float* dout, const float* out_fwd, int B, int N, int C, float* dqkv,
                             void* scratch, size_t scratch_bytes, void* stream);
/* Weight gradient of one convolution (autograd of nn.Conv2d inside `l_pix.backward()`, model/model.py:50-54), per op -- what
 * sr3_train_step runs per layer: dw[n][tap][c] = sum over output pixels of dy[m][n] * a_tap[m][c], with a = the convolution's
 * (virtual concat, optionally x2-upsampled, optionally GroupNorm-affine / SiLU-activated: act, ss as in sr3_conv_f32) input.
 * dy NHWC [B, Ho, Wo, Cout]; dw OHWI [Cout][ksize^2][C0 + C1], overwritten.  split != 0: the 3 x bf16 split kernel where it applies
 * (more than 64 channels on both sides; plan option wgrad_split), else the fp32-MFMA kernels.  scratch: the per-pixel-range slabs
 * (sr3_conv_wgrad_scratch_bytes), summed in double in a fixed order. */
size_t sr3_conv_wgrad_scratch_bytes(int B, int Hs, int Ws, int ups, int stride, int ksize, int C0, int C1, int Cout, int split);
int sr3_conv_wgrad_f32(const float* src0, int C0, const float* src1, int C1, int B, int Hs, int Ws, int ups, int stride, int ksize,
                       int Cout, const float* ss, int act, const float* dy, float* dw_ohwi, int split, void* scratch,
                       size_t scratch_bytes, void* stream);
/* noise-level / timestep embedding + MLP + all FiLM rows (unet.py:18-50,179-184): see sr3_common.h */
int sr3_film_embed_f32(int variant, int B, int inner, const float* level, const int64_t* timestep,
                       const float* freq, const float* w1, const float* b1, const float* w2, const float* b2,
                       const float* wf, const float* bf, int F, float* temb_scratch, float* film_out,
                       void* stream);
/* first conv (NCHW concat in -> NHWC) and final Block (NHWC -> NCHW) */
int sr3_conv_in_f32(const float* a_nchw, int Ca, const float* b_nchw, int Cb, int B, int H, int W,
                    const float* w_ohwi, const float* bias, int Cout, float* out_nhwc, void* stream);
int sr3_conv_out_f32(const float* x_nhwc, const float* ss, int B, int H, int W, int C, const float* w_ohwi,
                     const float* bias, int Cout, float* out_nchw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SR3_MI355X_H */
