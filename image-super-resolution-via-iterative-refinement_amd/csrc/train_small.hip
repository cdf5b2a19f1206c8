// Backward of the noise-level / timestep embedding MLP and of the per-block FiLM projections
// (PositionalEncoding -> Linear -> Swish -> Linear, FeatureWiseAffine / mlp Linear;
// sr3 unet.py:18-50,179-184, ddpm unet.py:19-34,81-84,165-170).  Tiny tensors (F = 8384 rows of 64):
// latency-bound, written for clarity.  Summation over the batch is in a fixed order.
#include "sr3_common.h"
#include "train.h"

namespace sr3 {

__device__ __forceinline__ float sig_e(float v) { return SR3_SIGMOID(v); }
__device__ __forceinline__ float dsilu_e(float v) { const float s = sig_e(v); return s * (1.0f + v * (1.0f - s)); }

// scratch layout per image b (floats): enc[inner] | hpre[4 inner] | tpre[inner] | e[inner] | de[inner]
__global__ __launch_bounds__(256) void k_embed_recompute(const EmbedBwdParams p) {
  extern __shared__ f32x4 smem_v[];
  float* enc = reinterpret_cast<float*>(smem_v);
  float* hid = enc + p.inner;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int half = p.inner / 2, hdim = 4 * p.inner;
  float* sc = p.scratch + (size_t)b * 8 * p.inner;
  const float lv = p.variant == 0 ? p.level[b] : (float)p.tstep[b];
  for (int k = tid; k < half; k += 256) {
    const float arg = lv * p.freq[k];
    enc[k] = sinf(arg);
    enc[half + k] = cosf(arg);
  }
  __syncthreads();
  for (int k = tid; k < p.inner; k += 256) sc[k] = enc[k];
  for (int j = tid; j < hdim; j += 256) {
    float s = p.b1[j];
    const float* wr = p.w1 + (size_t)j * p.inner;
    for (int k = 0; k < p.inner; ++k) s = fmaf(wr[k], enc[k], s);
    sc[p.inner + j] = s;
    hid[j] = s * sig_e(s);
  }
  __syncthreads();
  for (int j = tid; j < p.inner; j += 256) {
    float s = p.b2[j];
    const float* wr = p.w2 + (size_t)j * hdim;
    for (int k = 0; k < hdim; ++k) s = fmaf(wr[k], hid[k], s);
    sc[5 * p.inner + j] = s;
    sc[6 * p.inner + j] = p.variant == 1 ? s * sig_e(s) : s;
  }
}

// de[b][k] = sum_j dfilm[b][j] * wf[j][k]: F (8384 rows for the BASELINE network) is cut into FCH row chunks, one workgroup per
// (image, chunk) writes a partial, k_mlp_dtemb sums the chunks in order (round 6: one workgroup per image took 0.7 ms)
constexpr int FCH = 16;
__global__ __launch_bounds__(256) void k_film_bwd_input(const EmbedBwdParams p, float* __restrict__ parts) {
  __shared__ float red[256];
  const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
  const int k = tid % p.inner, part = tid / p.inner, nparts = 256 / p.inner;
  const int per = (p.F + FCH - 1) / FCH, j0 = ch * per, j1 = min(p.F, j0 + per);
  float s = 0.f;
  if (part < nparts)
    for (int j = j0 + part; j < j1; j += nparts) s = fmaf(p.dfilm[(size_t)b * p.F + j], p.wf[(size_t)j * p.inner + k], s);
  red[tid] = (part < nparts) ? s : 0.f;
  __syncthreads();
  if (tid < p.inner) {
    float a = 0.f;
    for (int q = 0; q < nparts; ++q) a += red[q * p.inner + tid];
    parts[((size_t)b * FCH + ch) * p.inner + tid] = a;
  }
}

// dwf[j][k] = sum_b dfilm[b][j] * e[b][k] ; dbf[j] = sum_b dfilm[b][j]
__global__ __launch_bounds__(256) void k_film_bwd_weights(const EmbedBwdParams p) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)p.F * p.inner) return;
  const int j = (int)(idx / p.inner), k = (int)(idx - (size_t)j * p.inner);
  float s = 0.f, sb = 0.f;
  for (int b = 0; b < p.B; ++b) {
    const float g = p.dfilm[(size_t)b * p.F + j];
    s = fmaf(g, p.scratch[(size_t)b * 8 * p.inner + 6 * p.inner + k], s);
    sb += g;
  }
  p.dwf[idx] = s;
  if (k == 0) p.dbf[j] = sb;
}

// MLP backward in three launches over many workgroups (round 6: the one-workgroup form took 1.4 ms of the step); every sum over the
// batch runs in one thread in image order, as before.  s2 (global): dtemb[B][inner] | dhpre[B][4 inner]
__global__ __launch_bounds__(256) void k_mlp_dtemb(const EmbedBwdParams p, const float* __restrict__ parts, float* __restrict__ s2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int inner = p.inner;
  if (i >= p.B * inner) return;
  const int b = i / inner, k = i - b * inner;
  float* sc = p.scratch + (size_t)b * 8 * inner;
  float d = 0.f;
  for (int c = 0; c < FCH; ++c) d += parts[((size_t)b * FCH + c) * inner + k];
  sc[7 * inner + k] = d;
  if (p.variant == 1) d *= dsilu_e(sc[5 * inner + k]);
  s2[i] = d;
}
// blocks [0, n_dw2): dw2[j][k] = sum_b dtemb[b][j] * h[b][k] (+ db2 from the first rows); then dhpre[b][k] = (sum_j dtemb[b][j] w2[j][k]) silu'(hpre)
__global__ __launch_bounds__(256) void k_mlp_bwd2(const EmbedBwdParams p, float* __restrict__ s2) {
  const int inner = p.inner, hdim = 4 * inner, B = p.B;
  const float* dtemb = s2;
  float* dhpre = s2 + (size_t)B * inner;
  const int n_dw2 = inner * hdim;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_dw2) {
    const int j = i / hdim, k = i - j * hdim;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      const float hp = p.scratch[(size_t)b * 8 * inner + inner + k];
      s = fmaf(dtemb[b * inner + j], hp * sig_e(hp), s);
    }
    p.dw2[i] = s;
    if (k == 0) {
      float sb = 0.f;
      for (int b = 0; b < B; ++b) sb += dtemb[b * inner + j];
      p.db2[j] = sb;
    }
    return;
  }
  i -= ((n_dw2 + 255) / 256) * 256;
  if (i >= 0 && i < B * hdim) {
    const int b = i / hdim, k = i - b * hdim;
    float s = 0.f;
    for (int j = 0; j < inner; ++j) s = fmaf(dtemb[b * inner + j], p.w2[(size_t)j * hdim + k], s);
    dhpre[i] = s * dsilu_e(p.scratch[(size_t)b * 8 * inner + inner + k]);
  }
}
// dw1[j][k] = sum_b dhpre[b][j] * enc[b][k], db1[j] = sum_b dhpre[b][j]
__global__ __launch_bounds__(256) void k_mlp_bwd1(const EmbedBwdParams p, const float* __restrict__ s2) {
  const int inner = p.inner, hdim = 4 * inner, B = p.B;
  const float* dhpre = s2 + (size_t)B * inner;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hdim * inner) return;
  const int j = i / inner, k = i - j * inner;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s = fmaf(dhpre[b * hdim + j], p.scratch[(size_t)b * 8 * inner + k], s);
  p.dw1[i] = s;
  if (k == 0) {
    float sb = 0.f;
    for (int b = 0; b < B; ++b) sb += dhpre[b * hdim + j];
    p.db1[j] = sb;
  }
}

int embed_backward(const EmbedBwdParams& p, hipStream_t st) {
  if (256 % p.inner) { set_error("embed_backward: inner must divide 256"); return SR3_E_UNSUPPORTED; }
  hipLaunchKernelGGL(k_embed_recompute, dim3(p.B), dim3(256), (size_t)5 * p.inner * sizeof(float), st, p);
  SR3_LAUNCH_CHECK("k_embed_recompute");
  float* s2 = p.scratch + (size_t)p.B * 8 * p.inner;     // dtemb | dhpre: B * 5 * inner floats
  float* parts = s2 + (size_t)p.B * 5 * p.inner;         // B * FCH * inner floats (caller sizes scratch as B * 29 * inner floats)
  hipLaunchKernelGGL(k_film_bwd_input, dim3(p.B, FCH), dim3(256), 0, st, p, parts);
  SR3_LAUNCH_CHECK("k_film_bwd_input");
  hipLaunchKernelGGL(k_mlp_dtemb, dim3((p.B * p.inner + 255) / 256), dim3(256), 0, st, p, parts, s2);
  SR3_LAUNCH_CHECK("k_mlp_dtemb");
  const size_t n = (size_t)p.F * p.inner;
  hipLaunchKernelGGL(k_film_bwd_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
  SR3_LAUNCH_CHECK("k_film_bwd_weights");
  const int hdim = 4 * p.inner;
  hipLaunchKernelGGL(k_mlp_bwd2, dim3((p.inner * hdim + 255) / 256 + (p.B * hdim + 255) / 256), dim3(256), 0, st, p, s2);
  SR3_LAUNCH_CHECK("k_mlp_bwd2");
  hipLaunchKernelGGL(k_mlp_bwd1, dim3((hdim * p.inner + 255) / 256), dim3(256), 0, st, p, s2);
  SR3_LAUNCH_CHECK("k_mlp_bwd1");
  return SR3_OK;
}

}  // namespace sr3
