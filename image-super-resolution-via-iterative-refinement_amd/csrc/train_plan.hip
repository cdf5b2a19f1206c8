// Training step driver: q_sample -> UNet forward (train plan) -> L1 loss -> backward walk -> gradient
// arena.  Replaces the reference's autograd step `l_pix = netG(data); l_pix.backward()`
// (model/model.py:50-54 over p_losses, model/sr3_modules/diffusion.py:221-246 /
// model/ddpm_modules/diffusion.py:278-294).  The backward is a reverse walk over the records the
// train plan kept for every conv / attention of the forward:
//   data gradient  = the same MFMA conv kernels run on dOut with flipped-transposed weights
//                    (zero insertion for the stride-2 convs, 2x2 sum for the nearest upsample),
//   weight gradient = wgrad.hip (activated input recomputed on the fly),
//   GroupNorm+SiLU  = reduce / fold / apply (train_kernels.hip), routed through the concat views,
//   attention       = attention_bwd.hip,
//   bias / FiLM     = per-channel sums of dOut (partial sums, fixed order).
// Every parameter gradient is written exactly once (no accumulation across ops), activation
// gradients live in a mirror of the activation arena and are accumulated in stream order.
#include <string.h>

#include <vector>

#include "plan_internal.h"
#include "train.h"

namespace sr3 {

namespace {
struct TrainCtx {
  sr3_plan* P;
  char* ws;
  const float* params;
  float* grads;
  int B;
  hipStream_t st;
  const float* act(int h) const { return reinterpret_cast<const float*>(ws + P->ttens[h].off); }
  float* grad(int h) const { return reinterpret_cast<float*>(ws + P->t_act_bytes + P->ttens[h].off); }
  template <typename T> T* at(size_t off) const { return reinterpret_cast<T*>(ws + off); }
};

// dOut [B,H,W,Cg] --(conv with flipped-transposed weights)--> dA [B,H,W,Cin]
int dgrad_conv(const TrainCtx& X, const float* g, int Cg, int H, int W, int ksize, const float* w, int Cout_w, int Cin,
               float* dA) {
  float* wt = X.at<float>(X.P->t_wt_off);
  int rc = w_flip_transpose(w, Cout_w, ksize * ksize, Cin, Cg, wt, X.st);
  if (rc) return rc;
  ConvParams c;
  memset(&c, 0, sizeof(c));
  c.src0 = g; c.C0 = Cg; c.B = X.B; c.Hs = H; c.Ws = W; c.stride = 1; c.ksize = ksize; c.Ho = H; c.Wo = W;
  c.Cout = Cin; c.w = wt; c.out = dA; c.ksplit = 1;
  // 3x3: Winograd F(2x2,3x3) on the flipped-transposed filters where the geometry fits (the data gradient has neither a
  // prologue nor dropout, so every 3x3 stride-1 / zero-inserted stride-2 layer with H, W multiples of 16 qualifies)
  WinoGeom wg;
  const bool split = X.P->wino_split != 0;            // (the one-image tile: the 3 x bf16 split instantiation, as the forward)
  if (X.P->winograd && ksize == 3 && wino_geometry(c, &wg) &&
      wino_weight_floats(Cin, Cg, split && wg.NB == 1) * sizeof(float) <= X.P->t_wu_bytes) {
    float* wu = X.at<float>(X.P->t_wu_off);
    c.wino_split = (split && wg.NB == 1) ? (X.P->wino2 ? 2 : 1) : 0;      // (2: the 8 x 16 tile of conv3x3_wino2.hip)
    rc = wino_transform_weights(wt, Cin, Cg, wu, X.st, c.wino_split != 0);
    if (rc) return rc;
    c.wino_u = wu;
    return conv_forward(c, 11, 0, X.at<float>(X.P->t_scratch_off), X.P->t_scratch_bytes, X.st);
  }
  // 1x1 / 8x8 data gradients on the im2col kernel: its 3 x bf16 split instantiation, with the forward Builder's exclusion (9-tap
  // layers producing <= 64 channels stay on the fp32 MFMA: plan.hip, Builder::conv) -- build_train sizes the scratch with the same rule
  c.igemm_split = (X.P->gemm_split && !(ksize == 3 && c.Cout <= 64)) ? 1 : 0;
  // 1x1: the plain GEMM kernel where it fits (plan option gemm2; gemm1x1.hip), its pre-split weights derived from the transposed filters
  // into the region the Winograd data gradients use for theirs
  if (c.igemm_split && X.P->gemm2 && gemm1x1_fits(c, 2) && (!(c.Cout & 127) || X.P->gemm_n64) && igemm_wsplit_floats(Cin, 1, Cg) * sizeof(float) <= X.P->t_wu_bytes) {
    float* ws_ = X.at<float>(X.P->t_wu_off);
    rc = igemm_split_weights(wt, Cin, 1, Cg, ws_, X.st);
    if (rc) return rc;
    c.w_split = ws_;
    int t = 22, ks = 0;
    conv_pick(c, t, ks);
    if ((size_t)ks * X.B * H * W * Cin * sizeof(float) <= X.P->t_scratch_bytes || ks == 1)
      return conv_forward(c, 22, ks, X.at<float>(X.P->t_scratch_off), X.P->t_scratch_bytes, X.st);
    c.w_split = nullptr;
  }
  return conv_forward(c, 0, 0, X.at<float>(X.P->t_scratch_off), X.P->t_scratch_bytes, X.st);
}

int wgrad_call(const TrainCtx& X, ConvParams c, const float* dy, float* dw) {
  WgradParams wp;
  c.wgrad_split = X.P->wgrad_split;
  wp.c = c;
  wp.dy = dy;
  wp.dw = dw;
  wp.slabs = X.at<float>(X.P->t_slab_off);
  wgrad_slab_bytes(c, &wp.msplit);
  return conv_wgrad(wp, X.st);
}
}  // namespace

int run_train(sr3_plan* P, const float* hr, const float* cond, int cond_channels, const float* z, const float* q_ca,
              const float* q_cb, const float* level, const int64_t* tstep, const float* freq, const float* params,
              float* grads, char* ws, float* loss_out, float grad_scale, int B, hipStream_t st, float dropout_p,
              unsigned seed, int n_marks, const size_t* marks, void* const* mark_events) {
  const sr3_unet_desc& d = P->d;
  const int S = d.image_size, G = d.norm_groups;
  const int xc = d.in_channel - cond_channels;
  TrainCtx X{P, ws, params, grads, B, st};
  int rc;
  // ---- q_sample + forward ----
  float* x_noisy = X.at<float>(P->t_xnoisy_off);
  float* eps = X.at<float>(P->t_eps_off);
  rc = q_sample(hr, z, q_ca, q_cb, B, xc * S * S, x_noisy, st);
  if (rc) return rc;
  Regions R;
  R.ops = &P->tops; R.stats_off = P->t_stats_off; R.ss_off = P->t_gn_off; R.mr_off = P->t_misc_off;
  R.temb_off = P->t_temb_off; R.film_off = P->t_film_off; R.scratch_off = P->t_scratch_off; R.scratch_bytes = P->t_scratch_bytes;
  DropCfg dc;
  dc.seed = seed;
  dropout_consts(dropout_p, &dc.thresh, &dc.scale);
  rc = run_forward(P, R, x_noisy, cond, cond_channels, level, tstep, freq, nullptr, nullptr, params, ws, eps, B, st, nullptr,
                   nullptr, &dc);
  if (rc) return rc;
  // ---- loss and its gradient (NHWC, channel dim padded to 4) ----
  float* geps = X.at<float>(P->t_geps_off);
  double* lparts = X.at<double>(P->t_dwtmp_off);
  rc = l1_loss_grad(z, eps, B, P->out_ch, S * S, 4, grad_scale, P->loss_l2 != 0, geps, lparts, loss_out, st);
  if (rc) return rc;
  // ---- the activation-gradient mirror is NOT zeroed (round 6: 1.5 ms per step): the first contribution to a tensor's gradient in this
  // walk is a plain store (`first` below), later ones accumulate in stream order; the FiLM gradient table is zeroed ----
  std::vector<char> seen(P->ttens.size(), 0);
  auto first = [&](int h) { if (h < 0) return false; const bool f = !seen[h]; seen[h] = 1; return f; };
  float* dfilm = X.at<float>(P->t_dfilm_off);
  SR3_HIP(hipMemsetAsync(dfilm, 0, (size_t)B * P->F * sizeof(float), st));

  float* dA = X.at<float>(P->t_dA_off);
  double* part = X.at<double>(P->t_part_off);
  double* gs = X.at<double>(P->t_gs_off);
  float* dwtmp = reinterpret_cast<float*>(ws + P->t_dwtmp_off + 4096 * sizeof(double));

  int next_mark = 0;
  for (int ri = (int)P->recs.size() - 1; ri >= 0; --ri) {
    // gradient-ready marks: every parameter at arena offset >= marks[k] has its gradient enqueued
    while (next_mark < n_marks && P->t_unproc_max[ri + 1] <= marks[next_mark]) {
      SR3_HIP(hipEventRecord(static_cast<hipEvent_t>(mark_events[next_mark]), st));
      ++next_mark;
    }
    const Rec& r = P->recs[ri];
    if (r.kind == R_CONV_OUT) {
      const Tensor& x0 = P->ttens[r.x0];
      const int C = x0.C;
      // bias (3 of the 4 padded columns are real; arena slots are 4-float aligned)
      rc = colsums(geps, B, S * S, 4, part, grads + r.bias, nullptr, 0, st);
      if (rc) return rc;
      rc = dgrad_conv(X, geps, 4, S, S, 3, params + r.w, P->out_ch, C, dA);
      if (rc) return rc;
      rc = act_bwd(dA, X.act(r.x0), nullptr, C, 0, B, S * S, X.at<float>(P->t_gn_off + r.ss_off),
                   X.at<float>(P->t_misc_off + r.mr_off), G, 2, params + r.gamma, part, gs, grads + r.gamma,
                   grads + r.beta, X.grad(r.x0), nullptr, st, 0u, 0u, 1.f, X.at<float>(P->t_a_off), !first(r.x0), true);
      if (rc) return rc;
      ConvParams c;
      memset(&c, 0, sizeof(c));
      float* abuf = X.at<float>(P->t_a_off);        // (the activated input, written by act_bwd's first pass)
      c.src0 = abuf; c.C0 = C; c.B = B; c.Hs = S; c.Ws = S; c.stride = 1; c.ksize = 3; c.Ho = S; c.Wo = S;
      c.Cout = 4;
      rc = wgrad_call(X, c, geps, dwtmp);
      if (rc) return rc;
      SR3_HIP(hipMemcpyAsync(grads + r.w, dwtmp, (size_t)P->out_ch * 9 * C * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else if (r.kind == R_ATTN) {
      const Tensor& o = P->ttens[r.o];
      // dK / dV through per-query-block slabs in the backward's scratch region, summed in block order: no atomics (round 6)
      if (!first(r.qkv)) { set_error("train: the qkv gradient has an earlier writer"); return SR3_E_UNSUPPORTED; }     // (attention_backward stores)
      rc = attention_backward(X.act(r.qkv), X.grad(r.o), X.act(r.o), B, o.H * o.W, o.C, X.grad(r.qkv), st,
                              X.at<float>(P->t_scratch_off), P->t_scratch_bytes);
      if (rc) return rc;
    } else if (r.kind == R_CONV_IN) {
      const Tensor& o = P->ttens[r.out];
      float* inpad = X.at<float>(P->t_inpad_off);
      const float* a = cond_channels > 0 ? cond : x_noisy;
      const int Ca = cond_channels > 0 ? cond_channels : xc;
      const float* b2 = cond_channels > 0 ? x_noisy : nullptr;
      const int Cb = cond_channels > 0 ? xc : 0;
      rc = nchw_to_nhwc_pad(a, Ca, b2, Cb, B, S * S, 8, inpad, st);
      if (rc) return rc;
      rc = colsums(X.grad(r.out), B, S * S, o.C, part, grads + r.bias, nullptr, 0, st);
      if (rc) return rc;
      ConvParams c;
      memset(&c, 0, sizeof(c));
      c.src0 = inpad; c.C0 = 8; c.B = B; c.Hs = S; c.Ws = S; c.stride = 1; c.ksize = 3; c.Ho = S; c.Wo = S; c.Cout = o.C;
      rc = wgrad_call(X, c, X.grad(r.out), dwtmp);
      if (rc) return rc;
      // compact [Cout][9][8] -> [Cout][9][in_channel]
      SR3_HIP(hipMemcpy2DAsync(grads + r.w, (size_t)d.in_channel * sizeof(float), dwtmp, 8 * sizeof(float),
                               (size_t)d.in_channel * sizeof(float), (size_t)o.C * 9, hipMemcpyDeviceToDevice, st));
    } else {
      const Tensor& x0 = P->ttens[r.x0];
      const Tensor& o = P->ttens[r.out];
      const int C0 = x0.C, C1 = r.x1 >= 0 ? P->ttens[r.x1].C : 0, Cin = C0 + C1;
      const int Ho = o.H, Wo = o.W, Cout = o.C;
      const float* g = X.grad(r.out);
      const float* x0p = X.act(r.x0);
      const float* x1p = r.x1 >= 0 ? X.act(r.x1) : nullptr;
      float* d0 = X.grad(r.x0);
      float* d1 = r.x1 >= 0 ? X.grad(r.x1) : nullptr;
      // 1. bias and FiLM gradients: column sums of dOut
      if (r.has_bias || r.film_row >= 0 || r.has_q) {
        rc = colsums(g, B, Ho * Wo, Cout, part, r.has_bias ? grads + r.bias : nullptr,
                     r.film_row >= 0 ? dfilm + r.film_row : nullptr, P->F, st);
        if (rc) return rc;
        if (r.has_q)      // res_conv bias sees the same sums
          SR3_HIP(hipMemcpyAsync(grads + r.qb, grads + r.bias, (size_t)Cout * sizeof(float), hipMemcpyDeviceToDevice, st));
      }
      // 2. identity residual
      if (r.r0 >= 0) {
        const bool a0 = !first(r.r0), a1 = !first(r.r1);
        rc = grad_route(g, P->ttens[r.r0].C, r.r1 >= 0 ? P->ttens[r.r1].C : 0, B, Ho, Wo, 0, X.grad(r.r0),
                        r.r1 >= 0 ? X.grad(r.r1) : nullptr, st, a0, a1);
        if (rc) return rc;
      }
      // 3. fused res_conv segment
      if (r.has_q) {
        const int Q0 = P->ttens[r.q0].C, Q1 = r.q1 >= 0 ? P->ttens[r.q1].C : 0;
        float* dq = X.at<float>(P->t_dq_off);
        rc = dgrad_conv(X, g, Cout, Ho, Wo, 1, params + r.qw, Cout, Q0 + Q1, dq);
        if (rc) return rc;
        const bool a0 = !first(r.q0), a1 = !first(r.q1);
        rc = grad_route(dq, Q0, Q1, B, Ho, Wo, 0, X.grad(r.q0), r.q1 >= 0 ? X.grad(r.q1) : nullptr, st, a0, a1);
        if (rc) return rc;
        ConvParams c;
        memset(&c, 0, sizeof(c));
        c.src0 = X.act(r.q0); c.src1 = r.q1 >= 0 ? X.act(r.q1) : nullptr; c.C0 = Q0; c.C1 = Q1; c.B = B; c.Hs = Ho; c.Ws = Wo;
        c.stride = 1; c.ksize = 1; c.Ho = Ho; c.Wo = Wo; c.Cout = Cout;
        rc = wgrad_call(X, c, g, grads + r.qw);
        if (rc) return rc;
      }
      // 4. main segment: data gradient
      const int Hi = x0.H << r.ups, Wi = x0.W << r.ups;
      const float* gsrc = g;
      if (r.stride == 2) {
        float* zb = X.at<float>(P->t_z_off);
        rc = zero_insert(g, B, Ho, Wo, Cout, zb, st);
        if (rc) return rc;
        gsrc = zb;
      }
      rc = dgrad_conv(X, gsrc, Cout, Hi, Wi, r.ksize, params + r.w, Cout, Cin, dA);
      if (rc) return rc;
      const bool dropped = r.has_drop && dc.thresh != 0;
      const unsigned lseed = drop_layer_seed(dc.seed, r.drop_key);
      const bool acc0 = !first(r.x0), acc1 = !first(r.x1);
      if (r.act) {
        rc = act_bwd(dA, x0p, x1p, C0, C1, B, x0.H * x0.W, X.at<float>(P->t_gn_off + r.ss_off),
                     X.at<float>(P->t_misc_off + r.mr_off), G, r.act, params + r.gamma, part, gs, grads + r.gamma,
                     grads + r.beta, d0, d1, st, lseed, dropped ? dc.thresh : 0u, dc.scale, X.at<float>(P->t_a_off), acc0, acc1);
      } else {
        rc = grad_route(dA, C0, C1, B, x0.H, x0.W, r.ups, d0, d1, st, acc0, acc1);
      }
      if (rc) return rc;
      // 5. weight gradient
      ConvParams c;
      memset(&c, 0, sizeof(c));
      c.B = B; c.Hs = x0.H; c.Ws = x0.W; c.ups = r.ups; c.stride = r.stride;
      c.ksize = r.ksize; c.Ho = Ho; c.Wo = Wo; c.Cout = Cout;
      if (r.act) {
        // the activated (and dropped) input is materialised once instead of being recomputed per tap: by act_bwd's first pass
        // above (round 6; it was a pass of its own over x, k_apply_act)
        float* abuf = X.at<float>(P->t_a_off);
        c.src0 = abuf; c.C0 = Cin; c.C1 = 0;
      } else {
        c.src0 = x0p; c.src1 = x1p; c.C0 = C0; c.C1 = C1;
      }
      rc = wgrad_call(X, c, g, grads + r.w);
      if (rc) return rc;
    }
  }
  while (next_mark < n_marks && P->t_unproc_max[0] <= marks[next_mark]) {
    SR3_HIP(hipEventRecord(static_cast<hipEvent_t>(mark_events[next_mark]), st));
    ++next_mark;
  }
  // ---- embedding MLP and FiLM projections ----
  EmbedBwdParams e;
  memset(&e, 0, sizeof(e));
  e.variant = d.variant; e.B = B; e.inner = d.inner_channel; e.F = P->F; e.level = level; e.tstep = tstep; e.freq = freq;
  e.w1 = params + P->emb_w1; e.b1 = params + P->emb_b1; e.w2 = params + P->emb_w2; e.b2 = params + P->emb_b2;
  e.wf = params + P->film_w; e.dfilm = dfilm;
  e.dw1 = grads + P->emb_w1; e.db1 = grads + P->emb_b1; e.dw2 = grads + P->emb_w2; e.db2 = grads + P->emb_b2;
  e.dwf = grads + P->film_w; e.dbf = grads + P->film_b;
  e.scratch = X.at<float>(P->t_embscr_off);
  rc = embed_backward(e, st);
  if (rc) return rc;
  while (next_mark < n_marks) {       // whatever is left becomes ready with the head block
    SR3_HIP(hipEventRecord(static_cast<hipEvent_t>(mark_events[next_mark]), st));
    ++next_mark;
  }
  return SR3_OK;
}

}  // namespace sr3

extern "C" {

size_t sr3_train_workspace_bytes(sr3_plan* plan, int batch, int cond_channels) {
  if (!plan) return 0;
  if (build_train(plan, batch, cond_channels)) return 0;
  return plan->t_ws_bytes;
}

int sr3_train_step(sr3_plan* plan, const float* hr_nchw, const float* cond_nchw, int cond_channels, const float* z_nchw,
                   const float* q_ca, const float* q_cb, const float* noise_level, const int64_t* timestep,
                   const float* freq, const float* params, float* grads, void* workspace, size_t workspace_bytes,
                   float* loss_sum_out, float grad_scale, float dropout_p, unsigned dropout_seed, int n_marks,
                   const size_t* mark_offsets, void* const* mark_events, int batch, void* stream) {
  if (!plan || !hr_nchw || !z_nchw || !q_ca || !q_cb || !freq || !params || !grads || !workspace || !loss_sum_out) {
    set_error("null argument");
    return SR3_E_BADARG;
  }
  if (!cond_nchw) cond_channels = 0;
  const int rc = build_train(plan, batch, cond_channels);
  if (rc) return rc;
  if (workspace_bytes < plan->t_ws_bytes) { set_error("train workspace too small: %zu < %zu", workspace_bytes, plan->t_ws_bytes); return SR3_E_NOMEM; }
  if (((uintptr_t)workspace & 255) || ((uintptr_t)params & 15) || ((uintptr_t)grads & 15)) { set_error("misaligned pointer"); return SR3_E_ALIGN; }
  if (plan->d.variant == SR3_VARIANT_SR3 && !noise_level) { set_error("SR3 variant needs noise_level"); return SR3_E_BADARG; }
  if (plan->d.variant == SR3_VARIANT_DDPM && !timestep) { set_error("DDPM variant needs timestep"); return SR3_E_BADARG; }
  if (dropout_p < 0.f || dropout_p >= 1.f) { set_error("dropout_p out of range"); return SR3_E_BADARG; }
  return run_train(plan, hr_nchw, cond_nchw, cond_channels, z_nchw, q_ca, q_cb, noise_level, timestep, freq, params, grads,
                   static_cast<char*>(workspace), loss_sum_out, grad_scale, batch, static_cast<hipStream_t>(stream), dropout_p,
                   dropout_seed, (mark_offsets && mark_events) ? n_marks : 0, mark_offsets, mark_events);
}

int sr3_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                  float beta2, float eps, int step, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || step < 1) { set_error("bad argument"); return SR3_E_BADARG; }
  return adam_step(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, static_cast<hipStream_t>(stream));
}

}  // extern "C"
