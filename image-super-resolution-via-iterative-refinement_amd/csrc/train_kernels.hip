// Backward-pass helpers of the training step (DDPM.optimize_parameters, model/model.py:48-58:
// p_losses -> backward -> Adam), HBM-bound: GroupNorm+SiLU backward (reduce / fold / apply),
// gradient routing through the virtual concat / residual / nearest-upsample / stride-2 views,
// weight re-layout for the data-gradient convolutions, bias / FiLM gradient sums, L1 loss and
// the fused Adam update.  The contractions (dgrad, wgrad, attention backward) live in
// conv3x3_halo.hip / conv_igemm.hip (reused with transformed weights), wgrad.hip and
// attention_bwd.hip.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "sr3_common.h"
#include "train.h"

namespace sr3 {

double meant_double(float f) {
  char buf[40];
  for (int prec = 1; prec <= 9; ++prec) {
    snprintf(buf, sizeof(buf), "%.*g", prec, (double)f);
    const double d = strtod(buf, nullptr);
    if ((float)d == f) return d;
  }
  return (double)f;
}

__device__ __forceinline__ float sigmoid_t(float v) { return SR3_SIGMOID(v); }

// ---------------------------------------------------------------------------------------------
// T1: activation backward + partial sums.  For the virtual concat x = (x0|x1) with u = x*scale+shift,
// a = silu(u) (act 2) or a = u (act 1):  du = dA * act'(u)  is written in place of dA, and per
// (image, channel) partials {sum du, sum du*xhat} go to part[B][T][C][2] (xhat = (x-mean)*rstd).
// Same geometry as k_chan_stats.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_act_bwd_reduce(float* __restrict__ dA, const float* __restrict__ x0,
                                                         const float* __restrict__ x1, int C0, int C1, int HW,
                                                         int LQ, int pix_per_block, const float* __restrict__ ss,
                                                         const float* __restrict__ mr, int groups, int act,
                                                         unsigned drop_seed, unsigned drop_thresh, float drop_scale,
                                                         double* __restrict__ part, float* __restrict__ aout) {
  __shared__ double red[256 * 8];
  const int tid = threadIdx.x;
  const int C = C0 + C1;
  const int nq = C >> 2;
  const int ql = tid % LQ, pl = tid / LQ, PP = 256 / LQ;
  const int q = blockIdx.y * LQ + ql;
  const int b = blockIdx.z;
  const int T = gridDim.x;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(HW, p0 + pix_per_block);
  const int cpg = C / groups;
  double s[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (q < nq) {
    const int c = q * 4;
    const bool second = c >= C0;
    const float* xs = second ? x1 : x0;
    const int Cs = second ? C1 : C0, cs = second ? c - C0 : c;
    float sc[4], sh[4], mu[4], rs[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sc[e] = ss[((size_t)b * C + c + e) * 2];
      sh[e] = ss[((size_t)b * C + c + e) * 2 + 1];
      const int g = (c + e) / cpg;
      mu[e] = mr[((size_t)b * groups + g) * 2];
      rs[e] = mr[((size_t)b * groups + g) * 2 + 1];
    }
    for (int p = p0 + pl; p < p1; p += PP) {
      const size_t pix = (size_t)b * HW + p;
      f32x4 g4 = *reinterpret_cast<const f32x4*>(dA + pix * C + c);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + pix * Cs + cs);
      f32x4 a4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float du = g4[e];
        float dm = 1.0f;
        if (drop_thresh != 0) { dm = drop_mask(drop_seed, (unsigned)(pix * C + c + e), drop_thresh, drop_scale); du *= dm; }
        const float u = fmaf(xv[e], sc[e], sh[e]);
        float av = u;
        if (act == 2) {
          const float sg = sigmoid_t(u);
          du *= sg * (1.0f + u * (1.0f - sg));
          av = u * sg;
        }
        if (drop_thresh != 0) av *= dm;
        a4[e] = av;
        g4[e] = du;
        const double xh = ((double)xv[e] - (double)mu[e]) * (double)rs[e];
        s[e] += (double)du;
        s2[e] += (double)du * xh;
      }
      *reinterpret_cast<f32x4*>(dA + pix * C + c) = g4;
      // the activated (and dropped) conv input, for the weight gradient that follows: the values k_apply_act writes, without its pass over x
      if (aout) *reinterpret_cast<f32x4*>(aout + pix * C + c) = a4;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = s[e]; red[tid * 8 + 4 + e] = s2[e]; }
  __syncthreads();
  if (pl == 0 && q < nq) {
    for (int k = 1; k < PP; ++k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[e] += red[(k * LQ + ql) * 8 + e]; s2[e] += red[(k * LQ + ql) * 8 + 4 + e]; }
    }
    double* o = part + (((size_t)b * T + blockIdx.x) * C + q * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = s[e]; o[2 * e + 1] = s2[e]; }
  }
}

// T2a: per (image, group): S1 = sum_c gamma_c * A_c, S2 = sum_c gamma_c * B_c  -> gs[B][G][2] (double)
__global__ __launch_bounds__(64) void k_gn_bwd_group(const double* __restrict__ part, int C, int T, int groups,
                                                      const float* __restrict__ gamma, double* __restrict__ gs) {
  const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
  const int cpg = C / groups;
  const int lane = threadIdx.x;
  double a = 0.0, bb = 0.0;
  for (int idx = lane; idx < cpg * T; idx += 64) {
    const int k = idx / T, t = idx - k * T;
    const int c = g * cpg + k;
    const double* q = part + (((size_t)b * T + t) * C + c) * 2;
    const double gm = (double)gamma[c];
    a += gm * q[0]; bb += gm * q[1];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m); bb += __shfl_xor(bb, m); }
  if (lane == 0) { gs[((size_t)b * groups + g) * 2] = a; gs[((size_t)b * groups + g) * 2 + 1] = bb; }
}

// T2b / T8: per channel: out0[c] (+)= sum_{b,t} part[..][c][0] ; out1[c] (+)= sum part[..][c][1]  (either may be null)
__global__ __launch_bounds__(64) void k_part_colsum(const double* __restrict__ part, int B, int C, int T,
                                                     float* __restrict__ out0, float* __restrict__ out1) {
  const int c = blockIdx.x;
  const int lane = threadIdx.x;
  double a = 0.0, bb = 0.0;
  for (int idx = lane; idx < B * T; idx += 64) {
    const double* q = part + ((size_t)idx * C + c) * 2;
    a += q[0]; bb += q[1];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m); bb += __shfl_xor(bb, m); }
  if (lane == 0) {
    if (out0) out0[c] = (float)a;
    if (out1) out1[c] = (float)bb;
  }
}

// per (image, channel): out[b*stride + c] = sum_t part[b][t][c][0]   (FiLM gradient rows)
__global__ __launch_bounds__(256) void k_part_imgsum(const double* __restrict__ part, int B, int C, int T,
                                                      float* __restrict__ out, int stride) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * C) return;
  const int b = idx / C, c = idx - b * C;
  double a = 0.0;
  for (int t = 0; t < T; ++t) a += part[(((size_t)b * T + t) * C + c) * 2];
  out[(size_t)b * stride + c] = (float)a;
}

// T3: dx(src) += rstd * (gamma*du - (S1 + xhat*S2)/n), routed to the two concat sources.
// The two terms cancel heavily (GroupNorm's backward projects the x-hat and the constant component out of du), so the
// per-(image, group) coefficients stay in double and the expression is evaluated in double and rounded once -- what
// torch's CPU GroupNorm backward does (acc_type<float> = double for its c1 / c2 / c3 coefficients).  With fp32
// coefficients the rounding of S2 is a perturbation along x-hat that is coherent over a whole (image, group) and is
// amplified by sqrt(B H W) in the weight gradient of the conv that produced x (measured 1.2e-4 normwise at B = 64).
__global__ __launch_bounds__(256) void k_gn_bwd_apply(const float* __restrict__ du, const float* __restrict__ x0,
                                                       const float* __restrict__ x1, int C0, int C1, int HW,
                                                       const float* __restrict__ mr, const double* __restrict__ gs,
                                                       int groups, const float* __restrict__ gamma,
                                                       float* __restrict__ dx0, float* __restrict__ dx1,
                                                       size_t total4, int acc0, int acc1) {
  const int C = C0 + C1;
  const int nq = C >> 2;
  const int cpg = C / groups;
  const double inv_n = 1.0 / ((double)HW * (double)cpg);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / nq;
    const int c = (int)(i - pix * nq) * 4;
    const int b = (int)(pix / HW);
    const bool second = c >= C0;
    const float* xs = second ? x1 : x0;
    float* ds = second ? dx1 : dx0;
    const int Cs = second ? C1 : C0, cs = second ? c - C0 : c;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(du + pix * C + c);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + pix * Cs + cs);
    // (acc == 0: this is the first contribution to that gradient tensor in the backward walk -- a plain store, the mirror is not zeroed)
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (second ? acc1 : acc0) o = *reinterpret_cast<const f32x4*>(ds + pix * Cs + cs);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int g = (c + e) / cpg;
      const double mu = (double)mr[((size_t)b * groups + g) * 2], rs = (double)mr[((size_t)b * groups + g) * 2 + 1];
      const double S1 = gs[((size_t)b * groups + g) * 2], S2 = gs[((size_t)b * groups + g) * 2 + 1];
      const double xh = ((double)xv[e] - mu) * rs;
      o[e] += (float)(rs * ((double)gamma[c + e] * (double)g4[e] - (S1 + xh * S2) * inv_n));
    }
    *reinterpret_cast<f32x4*>(ds + pix * Cs + cs) = o;
  }
}

// T4: route a gradient over the virtual concat to its sources: dst(src) += g[.., c]; `ups` sums the
// 2x2 children of every source pixel (backward of the nearest x2 upsample, unet.py:61).
__global__ __launch_bounds__(256) void k_grad_route(const float* __restrict__ g, int C0, int C1, int B, int Hs, int Ws,
                                                     int ups, float* __restrict__ d0, float* __restrict__ d1,
                                                     size_t total4, int acc0, int acc1) {
  const int C = C0 + C1;
  const int nq = C >> 2;
  const int Wg = Ws << ups;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / nq;                  // source pixel (b, y, x)
    const int c = (int)(i - pix * nq) * 4;
    const bool second = c >= C0;
    float* ds = second ? d1 : d0;
    const int Cs = second ? C1 : C0, cs = second ? c - C0 : c;
    f32x4 acc;
    if (ups) {
      const int x = (int)(pix % Ws);
      const size_t by = pix / Ws;               // b*Hs + y
      const int y = (int)(by % Hs);
      const size_t bb = by / Hs;
      const size_t base = ((bb * (Hs * 2) + 2 * y) * Wg + 2 * x);
      acc = *reinterpret_cast<const f32x4*>(g + base * C + c);
      acc += *reinterpret_cast<const f32x4*>(g + (base + 1) * C + c);
      acc += *reinterpret_cast<const f32x4*>(g + (base + Wg) * C + c);
      acc += *reinterpret_cast<const f32x4*>(g + (base + Wg + 1) * C + c);
    } else {
      acc = *reinterpret_cast<const f32x4*>(g + pix * C + c);
    }
    f32x4 o = acc;
    if (second ? acc1 : acc0) o += *reinterpret_cast<const f32x4*>(ds + pix * Cs + cs);
    *reinterpret_cast<f32x4*>(ds + pix * Cs + cs) = o;
  }
}

// T5: zero insertion for the stride-2 data gradient: z[b][2oh][2ow] = g[b][oh][ow], rest 0 (z pre-zeroed)
__global__ __launch_bounds__(256) void k_zero_insert(const float* __restrict__ g, int C, int Ho, int Wo,
                                                      float* __restrict__ z, size_t total4) {
  const int nq = C >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / nq;
    const int c = (int)(i - pix * nq) * 4;
    const int ow = (int)(pix % Wo);
    const size_t bo = pix / Wo;
    const int oh = (int)(bo % Ho);
    const size_t b = bo / Ho;
    const size_t zp = (b * (2 * Ho) + 2 * oh) * (2 * Wo) + 2 * ow;
    *reinterpret_cast<f32x4*>(z + zp * C + c) = *reinterpret_cast<const f32x4*>(g + pix * C + c);
  }
}

// T6: weights of the data-gradient conv: wt[c][taps-1-tap][n] = w[n][tap][c]   (w: [Cout][taps][Cin])
// CoutP >= Cout pads the new K dimension with zeros (Cout = 3 -> 4 for the output Block).
__global__ __launch_bounds__(256) void k_w_flip_transpose(const float* __restrict__ w, int Cout, int taps, int Cin,
                                                           int CoutP, float* __restrict__ wt) {
  const size_t total = (size_t)Cin * taps * CoutP;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % CoutP);
    const size_t r = i / CoutP;
    const int tp = (int)(r % taps);
    const int c = (int)(r / taps);
    wt[i] = n < Cout ? w[((size_t)n * taps + (taps - 1 - tp)) * Cin + c] : 0.f;
  }
}

// T11: the pixel loss of set_loss (diffusion.py:84-90,245) and its gradient w.r.t. eps_hat, written NHWC with the
// channel dim padded to CP.  L1 (nn.L1Loss(reduction='sum'), what define_G configures): loss_part[block] = sum |z - e|,
// g = -sign(z - e) * scale.  L2 (nn.MSELoss(reduction='sum'), loss_type 'l2'): sum (z - e)^2, g = -2 (z - e) * scale.
template <bool L2>
__global__ __launch_bounds__(256) void k_l1_loss_grad(const float* __restrict__ z, const float* __restrict__ e, int B,
                                                       int Cc, int HW, int CP, float scale, float* __restrict__ g_nhwc,
                                                       double* __restrict__ loss_part) {
  __shared__ double red[256];
  double s = 0.0;
  const size_t total = (size_t)B * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / HW, p = i - b * HW;
    for (int c = 0; c < CP; ++c) {
      float gv = 0.f;
      if (c < Cc) {
        const size_t idx = (b * Cc + c) * HW + p;
        const float d = z[idx] - e[idx];
        if (L2) {
          s += (double)d * (double)d;
          gv = -2.f * d * scale;
        } else {
          s += (double)fabsf(d);
          gv = d > 0.f ? -scale : (d < 0.f ? scale : 0.f);
        }
      }
      g_nhwc[i * CP + c] = gv;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(64) void k_sum_parts(const double* __restrict__ part, int n, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += part[i];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if (threadIdx.x == 0) out[0] = (float)s;
}

// T12: fused Adam over the whole parameter arena (torch.optim.Adam defaults, model/model.py:39-40):
// m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, size_t n4, float w1, float b2, float w2, float eps,
                                               float step, float bc2_sqrt) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 pv = *reinterpret_cast<f32x4*>(p + i * 4);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 mv = *reinterpret_cast<f32x4*>(m + i * 4);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mv[e] = mv[e] + (gv[e] - mv[e]) * w1;                     // torch: exp_avg.lerp_(grad, 1 - beta1)
      vv[e] = vv[e] * b2 + w2 * gv[e] * gv[e];                  //        exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pv[e] = pv[e] - step * (mv[e] / denom);
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pv;
    *reinterpret_cast<f32x4*>(m + i * 4) = mv;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
  }
}

// NCHW (C <= 4) -> NHWC with the channel dim padded to CP (input of the first conv for its wgrad)
__global__ __launch_bounds__(256) void k_nchw_to_nhwc_pad(const float* __restrict__ a, int Ca, const float* __restrict__ b2,
                                                           int Cb, int HW, int CP, float* __restrict__ out, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t bi = i / HW, p = i - bi * HW;
    for (int c = 0; c < CP; ++c) {
      float v = 0.f;
      if (c < Ca) v = a[(bi * Ca + c) * HW + p];
      else if (c < Ca + Cb) v = b2[(bi * Cb + (c - Ca)) * HW + p];
      out[i * CP + c] = v;
    }
  }
}

// materialise the activated conv input a = dropout(act(x * scale + shift)) over the virtual concat
// (input of the weight-gradient GEMM, so that it needs no per-tap recomputation)
__global__ __launch_bounds__(256) void k_apply_act(const float* __restrict__ x0, const float* __restrict__ x1, int C0,
                                                    int C1, int HW, const float* __restrict__ ss, int act,
                                                    unsigned drop_seed, unsigned drop_thresh, float drop_scale,
                                                    float* __restrict__ out, size_t total4) {
  const int C = C0 + C1;
  const int nq = C >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / nq;
    const int c = (int)(i - pix * nq) * 4;
    const int b = (int)(pix / HW);
    const bool second = c >= C0;
    const float* xs = second ? x1 : x0;
    const int Cs = second ? C1 : C0, cs = second ? c - C0 : c;
    f32x4 v = *reinterpret_cast<const f32x4*>(xs + pix * Cs + cs);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(ss + ((size_t)b * C + c) * 2);
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(ss + ((size_t)b * C + c) * 2 + 4);
    v.x = fmaf(v.x, s0.x, s0.y); v.y = fmaf(v.y, s0.z, s0.w); v.z = fmaf(v.z, s1.x, s1.y); v.w = fmaf(v.w, s1.z, s1.w);
    if (act == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * sigmoid_t(v[e]);
    }
    if (drop_thresh != 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= drop_mask(drop_seed, (unsigned)(pix * C + c + e), drop_thresh, drop_scale);
    }
    *reinterpret_cast<f32x4*>(out + pix * C + c) = v;
  }
}

// ---- host wrappers --------------------------------------------------------------------------------
static void stats_geometry(int B, int HW, int C, int* LQ_, int* cblocks_, int* ppb_, int* slices_) {
  const int nq = C >> 2;
  int LQ = 1;
  while (LQ < nq && LQ < 64) LQ <<= 1;
  const int cblocks = (nq + LQ - 1) / LQ;
  const int PP = 256 / LQ;
  long base_blocks = (long)cblocks * B;
  long want = 2048 / (base_blocks > 0 ? base_blocks : 1);
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  long max_slices = (HW + PP - 1) / PP;
  if (want > max_slices) want = max_slices;
  int ppb = (int)((HW + want - 1) / want);
  ppb = ((ppb + PP - 1) / PP) * PP;
  *LQ_ = LQ; *cblocks_ = cblocks; *ppb_ = ppb; *slices_ = (HW + ppb - 1) / ppb;
}

static inline int ew_blocks(size_t n) { size_t b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b ? b : 1)); }

int act_bwd(float* dA, const float* x0, const float* x1, int C0, int C1, int B, int HW, const float* ss, const float* mr,
            int groups, int act, const float* gamma, double* part, double* gs, float* dgamma, float* dbeta, float* dx0,
            float* dx1, hipStream_t st, unsigned drop_seed, unsigned drop_thresh, float drop_scale, float* aout, bool acc0, bool acc1) {
  const int C = C0 + C1;
  if ((C0 & 3) || (C1 & 3)) { set_error("act_bwd: channels %% 4"); return SR3_E_UNSUPPORTED; }
  int LQ, cblocks, ppb, T;
  stats_geometry(B, HW, C, &LQ, &cblocks, &ppb, &T);
  hipLaunchKernelGGL(k_act_bwd_reduce, dim3(T, cblocks, B), dim3(256), 0, st, dA, x0, x1, C0, C1, HW, LQ, ppb, ss, mr,
                     groups, act, drop_seed, drop_thresh, drop_scale, part, aout);
  SR3_LAUNCH_CHECK("k_act_bwd_reduce");
  hipLaunchKernelGGL(k_gn_bwd_group, dim3(B * groups), dim3(64), 0, st, part, C, T, groups, gamma, gs);
  SR3_LAUNCH_CHECK("k_gn_bwd_group");
  hipLaunchKernelGGL(k_part_colsum, dim3(C), dim3(64), 0, st, part, B, C, T, dbeta, dgamma);
  SR3_LAUNCH_CHECK("k_part_colsum");
  const size_t total4 = (size_t)B * HW * (C >> 2);
  hipLaunchKernelGGL(k_gn_bwd_apply, dim3(ew_blocks(total4)), dim3(256), 0, st, dA, x0, x1, C0, C1, HW, mr, gs, groups,
                     gamma, dx0, dx1, total4, acc0 ? 1 : 0, acc1 ? 1 : 0);
  SR3_LAUNCH_CHECK("k_gn_bwd_apply");
  return SR3_OK;
}
size_t act_bwd_part_bytes(int B, int HW, int C) {
  int LQ, cb, ppb, T;
  stats_geometry(B, HW, C, &LQ, &cb, &ppb, &T);
  return (size_t)B * T * C * 2 * sizeof(double);
}

int apply_act(const float* x0, const float* x1, int C0, int C1, int B, int HW, const float* ss, int act, unsigned drop_seed,
              unsigned drop_thresh, float drop_scale, float* out, hipStream_t st) {
  const size_t total4 = (size_t)B * HW * ((C0 + C1) >> 2);
  hipLaunchKernelGGL(k_apply_act, dim3(ew_blocks(total4)), dim3(256), 0, st, x0, x1, C0, C1, HW, ss, act, drop_seed,
                     drop_thresh, drop_scale, out, total4);
  SR3_LAUNCH_CHECK("k_apply_act");
  return SR3_OK;
}

int grad_route(const float* g, int C0, int C1, int B, int Hs, int Ws, int ups, float* d0, float* d1, hipStream_t st, bool acc0, bool acc1) {
  const size_t total4 = (size_t)B * Hs * Ws * ((C0 + C1) >> 2);
  hipLaunchKernelGGL(k_grad_route, dim3(ew_blocks(total4)), dim3(256), 0, st, g, C0, C1, B, Hs, Ws, ups, d0, d1, total4, acc0 ? 1 : 0, acc1 ? 1 : 0);
  SR3_LAUNCH_CHECK("k_grad_route");
  return SR3_OK;
}
int zero_insert(const float* g, int B, int Ho, int Wo, int C, float* z, hipStream_t st) {
  SR3_HIP(hipMemsetAsync(z, 0, (size_t)B * 4 * Ho * Wo * C * sizeof(float), st));
  const size_t total4 = (size_t)B * Ho * Wo * (C >> 2);
  hipLaunchKernelGGL(k_zero_insert, dim3(ew_blocks(total4)), dim3(256), 0, st, g, C, Ho, Wo, z, total4);
  SR3_LAUNCH_CHECK("k_zero_insert");
  return SR3_OK;
}
int w_flip_transpose(const float* w, int Cout, int taps, int Cin, int CoutP, float* wt, hipStream_t st) {
  const size_t total = (size_t)Cin * taps * CoutP;
  hipLaunchKernelGGL(k_w_flip_transpose, dim3(ew_blocks(total)), dim3(256), 0, st, w, Cout, taps, Cin, CoutP, wt);
  SR3_LAUNCH_CHECK("k_w_flip_transpose");
  return SR3_OK;
}
// bias gradient (sum over images and pixels) and / or FiLM gradient rows (sum over pixels) of g [B,HW,C]
int colsums(const float* g, int B, int HW, int C, double* part, float* dbias, float* dfilm, int film_stride,
            hipStream_t st) {
  int rc = chan_stats(g, B, HW, C, part, st);
  if (rc) return rc;
  const int T = chan_stats_slices(B, HW, C);
  if (dbias) {
    hipLaunchKernelGGL(k_part_colsum, dim3(C), dim3(64), 0, st, part, B, C, T, dbias, (float*)nullptr);
    SR3_LAUNCH_CHECK("k_part_colsum");
  }
  if (dfilm) {
    hipLaunchKernelGGL(k_part_imgsum, dim3((B * C + 255) / 256), dim3(256), 0, st, part, B, C, T, dfilm, film_stride);
    SR3_LAUNCH_CHECK("k_part_imgsum");
  }
  return SR3_OK;
}
int l1_loss_grad(const float* z, const float* e, int B, int Cc, int HW, int CP, float scale, bool l2, float* g_nhwc,
                 double* loss_part, float* loss_out, hipStream_t st) {
  const int blocks = 256;
  if (l2) hipLaunchKernelGGL(k_l1_loss_grad<true>, dim3(blocks), dim3(256), 0, st, z, e, B, Cc, HW, CP, scale, g_nhwc, loss_part);
  else hipLaunchKernelGGL(k_l1_loss_grad<false>, dim3(blocks), dim3(256), 0, st, z, e, B, Cc, HW, CP, scale, g_nhwc, loss_part);
  SR3_LAUNCH_CHECK("k_l1_loss_grad");
  hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(64), 0, st, loss_part, blocks, loss_out);
  SR3_LAUNCH_CHECK("k_sum_parts");
  return SR3_OK;
}
int adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int step,
              hipStream_t st) {
  if (n & 3) { set_error("adam: n %% 4"); return SR3_E_BADARG; }
  // torch.optim.Adam evaluates 1 - beta, the bias corrections and the step size as Python floats (double) and rounds
  // each ONCE to fp32 when it meets the tensor (meant_double: the decimal behind the ABI's fp32 hyper-parameters).
  const double b1d = meant_double(b1), b2d = meant_double(b2), lrd = meant_double(lr);
  const double bc1 = 1.0 - pow(b1d, (double)step);
  const double bc2 = 1.0 - pow(b2d, (double)step);
  const float step_size = (float)(lrd / bc1);
  const float bc2s = (float)sqrt(bc2);
  hipLaunchKernelGGL(k_adam, dim3(ew_blocks(n / 4)), dim3(256), 0, st, p, g, m, v, n / 4, (float)(1.0 - b1d), (float)b2d,
                     (float)(1.0 - b2d), (float)meant_double(eps), step_size, bc2s);
  SR3_LAUNCH_CHECK("k_adam");
  return SR3_OK;
}
int nchw_to_nhwc_pad(const float* a, int Ca, const float* b, int Cb, int B, int HW, int CP, float* out, hipStream_t st) {
  const size_t total = (size_t)B * HW;
  hipLaunchKernelGGL(k_nchw_to_nhwc_pad, dim3(ew_blocks(total)), dim3(256), 0, st, a, Ca, b, Cb, HW, CP, out, total);
  SR3_LAUNCH_CHECK("k_nchw_to_nhwc_pad");
  return SR3_OK;
}

}  // namespace sr3
