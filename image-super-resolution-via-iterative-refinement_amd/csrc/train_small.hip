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

// de[b][k] = sum_j dfilm[b][j] * wf[j][k]
__global__ __launch_bounds__(256) void k_film_bwd_input(const EmbedBwdParams p) {
  __shared__ float red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int k = tid % p.inner, part = tid / p.inner, nparts = 256 / p.inner;
  float s = 0.f;
  if (part < nparts)
    for (int j = part; j < p.F; j += nparts) s = fmaf(p.dfilm[(size_t)b * p.F + j], p.wf[(size_t)j * p.inner + k], s);
  red[tid] = (part < nparts) ? s : 0.f;
  __syncthreads();
  if (tid < p.inner) {
    float a = 0.f;
    for (int q = 0; q < nparts; ++q) a += red[q * p.inner + tid];
    p.scratch[(size_t)b * 8 * p.inner + 7 * p.inner + tid] = a;
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

// MLP backward, one block: uses scratch2 (global) for dtemb[B][inner] and dhpre[B][4 inner]
__global__ __launch_bounds__(256) void k_mlp_bwd(const EmbedBwdParams p, float* __restrict__ s2) {
  const int tid = threadIdx.x;
  const int inner = p.inner, hdim = 4 * inner, B = p.B;
  float* dtemb = s2;                       // [B][inner]
  float* dhpre = s2 + (size_t)B * inner;   // [B][hdim]
  for (int i = tid; i < B * inner; i += 256) {
    const int b = i / inner, k = i - b * inner;
    const float* sc = p.scratch + (size_t)b * 8 * inner;
    float d = sc[7 * inner + k];
    if (p.variant == 1) d *= dsilu_e(sc[5 * inner + k]);
    dtemb[i] = d;
  }
  __syncthreads();
  for (int j = tid; j < inner; j += 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dtemb[b * inner + j];
    p.db2[j] = s;
  }
  for (int i = tid; i < inner * hdim; i += 256) {        // dw2[j][k] = sum_b dtemb[b][j] * h[b][k]
    const int j = i / hdim, k = i - j * hdim;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      const float hp = p.scratch[(size_t)b * 8 * inner + inner + k];
      s = fmaf(dtemb[b * inner + j], hp * sig_e(hp), s);
    }
    p.dw2[i] = s;
  }
  for (int i = tid; i < B * hdim; i += 256) {            // dhpre[b][k] = (sum_j dtemb[b][j] w2[j][k]) * silu'(hpre)
    const int b = i / hdim, k = i - b * hdim;
    float s = 0.f;
    for (int j = 0; j < inner; ++j) s = fmaf(dtemb[b * inner + j], p.w2[(size_t)j * hdim + k], s);
    dhpre[i] = s * dsilu_e(p.scratch[(size_t)b * 8 * inner + inner + k]);
  }
  __syncthreads();
  for (int j = tid; j < hdim; j += 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dhpre[b * hdim + j];
    p.db1[j] = s;
  }
  for (int i = tid; i < hdim * inner; i += 256) {        // dw1[j][k] = sum_b dhpre[b][j] * enc[b][k]
    const int j = i / inner, k = i - j * inner;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s = fmaf(dhpre[b * hdim + j], p.scratch[(size_t)b * 8 * inner + k], s);
    p.dw1[i] = s;
  }
}

int embed_backward(const EmbedBwdParams& p, hipStream_t st) {
  if (256 % p.inner) { set_error("embed_backward: inner must divide 256"); return SR3_E_UNSUPPORTED; }
  hipLaunchKernelGGL(k_embed_recompute, dim3(p.B), dim3(256), (size_t)5 * p.inner * sizeof(float), st, p);
  SR3_LAUNCH_CHECK("k_embed_recompute");
  hipLaunchKernelGGL(k_film_bwd_input, dim3(p.B), dim3(256), 0, st, p);
  SR3_LAUNCH_CHECK("k_film_bwd_input");
  const size_t n = (size_t)p.F * p.inner;
  hipLaunchKernelGGL(k_film_bwd_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
  SR3_LAUNCH_CHECK("k_film_bwd_weights");
  float* s2 = p.scratch + (size_t)p.B * 8 * p.inner;     // caller sizes scratch as B * 13 * inner floats
  hipLaunchKernelGGL(k_mlp_bwd, dim3(1), dim3(256), 0, st, p, s2);
  SR3_LAUNCH_CHECK("k_mlp_bwd");
  return SR3_OK;
}

}  // namespace sr3
