// Single-head self-attention core of SelfAttention.forward (model/sr3_modules/unet.py:127-139):
//   S = Q K^T / sqrt(C) ; P = softmax_keys(S) ; O = P V          (n_head == 1, head dim == C)
// qkv is the NHWC output of the 1x1 qkv conv: [B][N][3C] with q | k | v along channels
// (the reference's `.view(b, 1, 3C, h, w).chunk(3, dim=2)`), O is [B][N][C].
//
// One workgroup (4 waves) owns 32 query rows of one image.  N <= 1024 tokens, so a full score
// strip S[32][N] lives in LDS (no online softmax): phase 1 fills it with exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32, each wave a 32-key block), phase 2 does the row softmax in place,
// phase 3 multiplies the strip with V (each wave a 32-channel block of a 128-channel panel).
#include "sr3_common.h"

namespace sr3 {

constexpr int AT_LDK = 36;    // Q/K staging row stride (32 + 4 pad floats)
constexpr int AT_LDV = 132;   // V staging row stride (128 + 4)

__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ qkv, int N, int C,
                                                    float* __restrict__ out) {
  extern __shared__ f32x4 smem_v[];
  float* smem = reinterpret_cast<float*>(smem_v);
  const int Npad = (N + 31) & ~31;
  const int LDS_S = Npad + 4;
  float* S = smem;                      // [32][LDS_S]
  float* stg = smem + 32 * LDS_S;       // staging union
  float* Qs = stg;                      // [32][AT_LDK]
  float* Ks = stg + 32 * AT_LDK;        // [128][AT_LDK]
  float* Vs = stg;                      // [32][AT_LDV]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int m0 = blockIdx.x * 32;
  const size_t rowstride = (size_t)3 * C;
  const float* base = qkv + (size_t)b * N * rowstride;
  const int kq = tid & 7, lrow = tid >> 3;        // loaders: 8 float4 per 32-channel row
  const int kh = (lane >> 5) * 4;
  const float inv_div = sqrtf((float)C);

  // ---------------- phase 1: S = Q K^T / sqrt(C) ----------------
  for (int kb = 0; kb < Npad; kb += 128) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool wave_active = (kb + wave * 32) < Npad;
    for (int c0 = 0; c0 < C; c0 += 32) {
      const int c = c0 + kq * 4;
      __syncthreads();
      {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int m = m0 + lrow;
        if (m < N && c < C) v = *reinterpret_cast<const f32x4*>(base + (size_t)m * rowstride + c);
        *reinterpret_cast<f32x4*>(&Qs[lrow * AT_LDK + kq * 4]) = v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int key = kb + lrow + 32 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (key < N && c < C) v = *reinterpret_cast<const f32x4*>(base + (size_t)key * rowstride + C + c);
        *reinterpret_cast<f32x4*>(&Ks[(lrow + 32 * i) * AT_LDK + kq * 4]) = v;
      }
      __syncthreads();
      if (wave_active) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(&Qs[(lane & 31) * AT_LDK + kk * 8 + kh]);
          const f32x4 k4 = *reinterpret_cast<const f32x4*>(&Ks[(wave * 32 + (lane & 31)) * AT_LDK + kk * 8 + kh]);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], k4[q], acc, 0, 0, 0);
        }
      }
    }
    if (wave_active) {
      const int key = kb + wave * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        S[row * LDS_S + key] = acc[r] / inv_div;
      }
    }
  }
  __syncthreads();

  // ---------------- phase 2: row softmax over the N valid keys ----------------
  {
    const int row = tid >> 3, sub = tid & 7;
    float* sr = S + row * LDS_S;
    float mx = -INFINITY;
    for (int k = sub; k < N; k += 8) mx = fmaxf(mx, sr[k]);
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    mx = fmaxf(mx, __shfl_xor(mx, 4));
    float sum = 0.f;
    for (int k = sub; k < N; k += 8) { const float e = expf(sr[k] - mx); sr[k] = e; sum += e; }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    sum += __shfl_xor(sum, 4);
    for (int k = sub; k < N; k += 8) sr[k] = sr[k] / sum;
    for (int k = N + sub; k < Npad; k += 8) sr[k] = 0.f;
  }
  __syncthreads();

  // ---------------- phase 3: O = P V ----------------
  for (int cp = 0; cp < C; cp += 128) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool wave_active = (cp + wave * 32) < C;
    for (int k0 = 0; k0 < Npad; k0 += 32) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kr = (tid >> 5) + 8 * i;         // key row within the chunk
        const int c = cp + (tid & 31) * 4;
        const int key = k0 + kr;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (key < N && c < C) v = *reinterpret_cast<const f32x4*>(base + (size_t)key * rowstride + 2 * C + c);
        *reinterpret_cast<f32x4*>(&Vs[kr * AT_LDV + (tid & 31) * 4]) = v;
      }
      __syncthreads();
      if (wave_active) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(&S[(lane & 31) * LDS_S + k0 + kk * 8 + kh]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float bv = Vs[(kk * 8 + kh + q) * AT_LDV + wave * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], bv, acc, 0, 0, 0);
          }
        }
      }
    }
    if (wave_active) {
      const int c = cp + wave * 32 + (lane & 31);
      if (c < C) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < N) out[((size_t)b * N + m) * C + c] = acc[r];
        }
      }
    }
  }
}

int attention_forward(const float* qkv, int B, int N, int C, float* out, hipStream_t st) {
  if (C & 3) { set_error("attention: C %% 4 != 0"); return SR3_E_UNSUPPORTED; }
  const int Npad = (N + 31) & ~31;
  const size_t stg = (size_t)((32 + 128) * AT_LDK > 32 * AT_LDV ? (32 + 128) * AT_LDK : 32 * AT_LDV);
  const size_t smem = ((size_t)32 * (Npad + 4) + stg) * sizeof(float);
  if (smem > 160 * 1024) { set_error("attention: N=%d does not fit the LDS score strip", N); return SR3_E_UNSUPPORTED; }
  static size_t attr_max = 0;
  if (smem > attr_max) {
    SR3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_max = smem;
  }
  hipLaunchKernelGGL(k_attention, dim3((N + 31) / 32, B), dim3(256), smem, st, qkv, N, C, out);
  SR3_LAUNCH_CHECK("k_attention");
  return SR3_OK;
}

}  // namespace sr3
