// Single-head self-attention core of SelfAttention.forward (model/sr3_modules/unet.py:127-139):
//   S = Q K^T / sqrt(C) ; P = softmax_keys(S) ; O = P V          (n_head == 1, head dim == C)
// qkv is the NHWC output of the 1x1 qkv conv: [B][N][3C] with q | k | v along channels
// (the reference's `.view(b, 1, 3C, h, w).chunk(3, dim=2)`), O is [B][N][C].
//
// One workgroup (4 waves) owns 32 query rows of one image.  N <= 1024 tokens, so a full score
// strip S[32][N] lives in LDS (no online softmax): phase 1 fills it with exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32, each wave a 32-key block), phase 2 does the row softmax in place,
// phase 3 multiplies the strip with V (each wave a 32-channel block of a 128-channel panel).
// Q/K/V tiles are register-prefetched one step ahead (unconditional loads, masks applied at the
// LDS write) and double-buffered in LDS when the strip leaves room (NSTAGE = 2).  When the grid
// would leave CUs idle, the channel panels of phase 3 are split over gridDim.z workgroups (each
// recomputes the cheap score strip).
#include <stdlib.h>

#include "sr3_common.h"

namespace sr3 {

constexpr int AT_LDK = 36;    // Q/K staging row stride (32 + 4 pad floats)
constexpr int AT_LDV = 132;   // V staging row stride (128 + 4)
constexpr int AT_QK_STAGE = (32 + 128) * AT_LDK;
constexpr int AT_V_STAGE = 32 * AT_LDV;
constexpr int AT_STAGE = AT_QK_STAGE > AT_V_STAGE ? AT_QK_STAGE : AT_V_STAGE;

template <int NSTAGE>
__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ qkv, int N, int C,
                                                    float* __restrict__ out) {
  extern __shared__ f32x4 smem_v[];
  float* smem = reinterpret_cast<float*>(smem_v);
  const int Npad = (N + 31) & ~31;
  const int LDS_S = Npad + 4;
  float* S = smem;                      // [32][LDS_S]
  float* stg = smem + 32 * LDS_S;       // NSTAGE staging buffers

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int m0 = blockIdx.x * 32;
  const int rowstride = 3 * C;
  const float* base = qkv + (size_t)b * N * rowstride;
  const int kq = tid & 7, lrow = tid >> 3;        // loaders: 8 float4 per 32-channel row
  const int kh = (lane >> 5) * 4;
  const float sqrt_c = sqrtf((float)C);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  // ---------------- phase 1: S = Q K^T / sqrt(C) ----------------
  {
    const int nc = (C + 31) / 32;
    const int nsteps = (Npad / 128 + ((Npad & 127) ? 1 : 0)) * nc;
    f32x4 rq, rk[4];
    bool qok, kok[4];
    auto load = [&](int s) {
      const int kb = (s / nc) * 128;
      const int c = (s % nc) * 32 + kq * 4;
      const bool cv = c < C;
      const int m = m0 + lrow;
      qok = cv && m < N;
      rq = *reinterpret_cast<const f32x4*>(base + (qok ? m * rowstride + c : 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int key = kb + lrow + 32 * i;
        kok[i] = cv && key < N;
        rk[i] = *reinterpret_cast<const f32x4*>(base + (kok[i] ? key * rowstride + C + c : 0));
      }
    };
    auto store = [&](int st) {
      float* Qs = stg + st * AT_STAGE;
      float* Ks = Qs + 32 * AT_LDK;
      *reinterpret_cast<f32x4*>(&Qs[lrow * AT_LDK + kq * 4]) = qok ? rq : zero;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(&Ks[(lrow + 32 * i) * AT_LDK + kq * 4]) = kok[i] ? rk[i] : zero;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    load(0);
    store(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      const int cur = (NSTAGE == 2) ? (s & 1) : 0;
      const bool more = s + 1 < nsteps;
      if (more) load(s + 1);
      const int kb = (s / nc) * 128;
      const bool wave_active = (kb + wave * 32) < Npad;
      if (wave_active) {
        const float* Qs = stg + cur * AT_STAGE;
        const float* Ks = Qs + 32 * AT_LDK;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(&Qs[(lane & 31) * AT_LDK + kk * 8 + kh]);
          const f32x4 k4 = *reinterpret_cast<const f32x4*>(&Ks[(wave * 32 + (lane & 31)) * AT_LDK + kk * 8 + kh]);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], k4[q], acc, 0, 0, 0);
        }
        if ((s % nc) == nc - 1) {           // last channel chunk of this key block: emit the scores
          const int key = kb + wave * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            S[row * LDS_S + key] = acc[r] / sqrt_c;
            acc[r] = 0.f;
          }
        }
      }
      if (NSTAGE == 1) __syncthreads();
      if (more) store((NSTAGE == 2) ? (cur ^ 1) : 0);
      __syncthreads();
    }
  }

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

  // ---------------- phase 3: O = P V  (this workgroup's share of the 128-channel panels) --------
  {
    const int npan = (C + 127) / 128;
    const int pan_per = (npan + gridDim.z - 1) / gridDim.z;
    const int pan0 = blockIdx.z * pan_per;
    const int pan1 = min(npan, pan0 + pan_per);
    const int nk = Npad / 32;
    const int nsteps = (pan1 - pan0) * nk;
    f32x4 rv[4];
    bool vok[4];
    auto load = [&](int s) {
      const int cp = (pan0 + s / nk) * 128;
      const int k0 = (s % nk) * 32;
      const int c = cp + (tid & 31) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int key = k0 + (tid >> 5) + 8 * i;
        vok[i] = key < N && c < C;
        rv[i] = *reinterpret_cast<const f32x4*>(base + (vok[i] ? key * rowstride + 2 * C + c : 0));
      }
    };
    auto store = [&](int st) {
      float* Vs = stg + st * AT_STAGE;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(&Vs[((tid >> 5) + 8 * i) * AT_LDV + (tid & 31) * 4]) = vok[i] ? rv[i] : zero;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (nsteps > 0) {
      load(0);
      store(0);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      const int cur = (NSTAGE == 2) ? (s & 1) : 0;
      const bool more = s + 1 < nsteps;
      if (more) load(s + 1);
      const int cp = (pan0 + s / nk) * 128;
      const int k0 = (s % nk) * 32;
      const bool wave_active = (cp + wave * 32) < C;
      if (wave_active) {
        const float* Vs = stg + cur * AT_STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(&S[(lane & 31) * LDS_S + k0 + kk * 8 + kh]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float bv = Vs[(kk * 8 + kh + q) * AT_LDV + wave * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], bv, acc, 0, 0, 0);
          }
        }
        if ((s % nk) == nk - 1) {           // last key chunk of this panel: write the output block
          const int c = cp + wave * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < N && c < C) out[((size_t)b * N + m) * C + c] = acc[r];
            acc[r] = 0.f;
          }
        }
      }
      if (NSTAGE == 1) __syncthreads();
      if (more) store((NSTAGE == 2) ? (cur ^ 1) : 0);
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 3: the same computation with NO operand staging -- every MFMA operand except the probabilities comes straight from
// global memory in fragment form, because on this part nothing overlaps the fp32 MFMA (profiles/archive/r03_mfma_overlap.txt), so
// the cheapest operand is the one that costs the fewest instructions:
//   phase 1  S = Q K^T / sqrt(C): a lane's A / B operand for 4 consecutive k-steps is 16 contiguous bytes of ITS query /
//            key row (qkv is channel-contiguous), i.e. one global_load_dwordx4; a key block belongs to exactly one wave, so
//            LDS would only add a write and a read per fragment.  KP key blocks per wave share the Q fragment.
//   phase 2  exact row softmax in the LDS strip (as above).
//   phase 3  O = P V: A = P from the strip (one ds_read_b128 per 4 k-steps); the wave's TN 32-channel MFMA tiles take the
//            INTERLEAVED channels c0 + TN n + t (tile t, column n), so one TN-float load per lane and k-step feeds all TN
//            tiles and the accumulators of a row come out as TN consecutive channels (vector stores).
// No barrier inside either loop; D = 4 operand groups are in flight per wave (the grid is one workgroup per CU, i.e. one wave
// per SIMD: the L2 latency has to be covered inside the wave).  Workgroups are numbered so that all workgroups of an image
// land on ONE XCD (round-robin dispatch: linear id mod 8): its K / V are then fetched into one L2 instead of eight.
// The k order of both contractions is the one of k_attention (bitwise equal results).
// Shapes: N % 32 == 0, N <= 1024, C % 128 == 0, (C / zsplit) % (128 TN) == 0; the rest stays on k_attention.
// ---------------------------------------------------------------------------------------------------------------
template <int TN> struct AtVec;
template <> struct AtVec<1> { typedef float type; };
template <> struct AtVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct AtVec<4> { typedef f32x4 type; };
template <int TN> __device__ __forceinline__ float at_elem(const typename AtVec<TN>::type& v, int t) { return v[t]; }
template <> __device__ __forceinline__ float at_elem<1>(const float& v, int) { return v; }

// SPLIT (round 5; plan option attn_split): both contractions on v_mfma_f32_32x32x16_bf16 with every fp32 operand as three bf16
// terms and six products per fp32 product, fp32 accumulation -- the arithmetic of the SPLIT conv kernels (fp32-class results,
// gated against float64 in tests/).  A lane's operand is then 8 consecutive k of its row: 32 contiguous bytes of its query / key
// row (phase 1), 8 probabilities of the strip and 8 keys' worth of its TN channels (phase 3), split in registers; D = 2 groups
// of 16 k in flight.
template <int KP, int TN, bool SPLIT>
__global__ __launch_bounds__(256) void k_attention_v2(const float* __restrict__ qkv, int B, int N, int C, int zsplit,
                                                       float* __restrict__ out) {
  extern __shared__ f32x4 smem_v[];
  float* S = reinterpret_cast<float*>(smem_v);              // [32][N + 4]
  const int LDS_S = N + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (query block, channel split, image) of this workgroup; B % 8 == 0: image = xcd + 8 * (...), xcd = linear id mod 8
  const int qblocks = N >> 5, per_img = qblocks * zsplit;
  int b, r;
  if ((B & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    b = xcd + 8 * (j / per_img);
    r = j % per_img;
  } else {
    b = blockIdx.x / per_img;
    r = blockIdx.x % per_img;
  }
  const int m0 = (r % qblocks) * 32, zi = r / qblocks;
  const int rowstride = 3 * C;
  const float* base = qkv + (size_t)b * N * rowstride;
  const int ln = lane & 31, kh = (lane >> 5) * 4;
  const float sqrt_c = sqrtf((float)C);
  constexpr int D = 4;                                      // operand groups in flight

  // ---------------- phase 1 ----------------
  if constexpr (SPLIT) {
    const int KB = N >> 5, G = C >> 4;                      // groups of 16 channels = one bf16 MFMA k-step; G % 2 == 0 (C % 128 == 0)
    const int k8 = (lane >> 5) * 8;
    const float* qrow = base + (size_t)(m0 + ln) * rowstride + k8;
    constexpr int DS = 2;
    for (int kb0 = wave * KP; kb0 < KB; kb0 += 4 * KP) {
      const float* krow[KP];
#pragma unroll
      for (int p = 0; p < KP; ++p) krow[p] = base + (size_t)(min(kb0 + p, KB - 1) * 32 + ln) * rowstride + C + k8;
      f32x16 acc[KP];
#pragma unroll
      for (int p = 0; p < KP; ++p)
#pragma unroll
        for (int r_ = 0; r_ < 16; ++r_) acc[p][r_] = 0.f;
      f32x4 a[DS][2], k4[DS][KP][2];
#pragma unroll
      for (int d = 0; d < DS; ++d)
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
          a[d][hlf] = *reinterpret_cast<const f32x4*>(qrow + d * 16 + hlf * 4);
#pragma unroll
          for (int p = 0; p < KP; ++p) k4[d][p][hlf] = *reinterpret_cast<const f32x4*>(krow[p] + d * 16 + hlf * 4);
        }
      for (int g0 = 0; g0 < G; g0 += DS) {
#pragma unroll
        for (int d = 0; d < DS; ++d) {
          bf16x8 qa[3];
          split3x8(a[d][0], a[d][1], qa[0], qa[1], qa[2]);
#pragma unroll
          for (int p = 0; p < KP; ++p) {
            bf16x8 kb[3];
            split3x8(k4[d][p][0], k4[d][p][1], kb[0], kb[1], kb[2]);
            mfma_split6(qa, kb, acc[p]);
          }
          const int gn = min(g0 + d + DS, G - 1) * 16;        // (the tail re-fetches the last group: the loads stay unconditional)
#pragma unroll
          for (int hlf = 0; hlf < 2; ++hlf) {
            a[d][hlf] = *reinterpret_cast<const f32x4*>(qrow + gn + hlf * 4);
#pragma unroll
            for (int p = 0; p < KP; ++p) k4[d][p][hlf] = *reinterpret_cast<const f32x4*>(krow[p] + gn + hlf * 4);
          }
        }
      }
#pragma unroll
      for (int p = 0; p < KP; ++p) {
        if (kb0 + p < KB) {
          const int key = (kb0 + p) * 32 + ln;
#pragma unroll
          for (int r_ = 0; r_ < 16; ++r_) S[((r_ & 3) + 8 * (r_ >> 2) + kh) * LDS_S + key] = acc[p][r_] / sqrt_c;
        }
      }
    }
  } else {
    const int KB = N >> 5, G = C >> 3;                      // G % D == 0 (C % 32 == 0: host)
    const float* qrow = base + (size_t)(m0 + ln) * rowstride + kh;
    for (int kb0 = wave * KP; kb0 < KB; kb0 += 4 * KP) {
      const float* krow[KP];
#pragma unroll
      for (int p = 0; p < KP; ++p) krow[p] = base + (size_t)(min(kb0 + p, KB - 1) * 32 + ln) * rowstride + C + kh;
      f32x16 acc[KP];
#pragma unroll
      for (int p = 0; p < KP; ++p)
#pragma unroll
        for (int r_ = 0; r_ < 16; ++r_) acc[p][r_] = 0.f;
      f32x4 a[D], k4[D][KP];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        a[d] = *reinterpret_cast<const f32x4*>(qrow + d * 8);
#pragma unroll
        for (int p = 0; p < KP; ++p) k4[d][p] = *reinterpret_cast<const f32x4*>(krow[p] + d * 8);
      }
      for (int g0 = 0; g0 < G; g0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int p = 0; p < KP; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][q], k4[d][p][q], acc[p], 0, 0, 0);
          const int gn = min(g0 + d + D, G - 1) * 8;          // (the tail re-fetches the last group: the loads stay unconditional)
          a[d] = *reinterpret_cast<const f32x4*>(qrow + gn);
#pragma unroll
          for (int p = 0; p < KP; ++p) k4[d][p] = *reinterpret_cast<const f32x4*>(krow[p] + gn);
        }
      }
#pragma unroll
      for (int p = 0; p < KP; ++p) {
        if (kb0 + p < KB) {
          const int key = (kb0 + p) * 32 + ln;
#pragma unroll
          for (int r_ = 0; r_ < 16; ++r_) S[((r_ & 3) + 8 * (r_ >> 2) + kh) * LDS_S + key] = acc[p][r_] / sqrt_c;
        }
      }
    }
  }
  __syncthreads();

  // ---------------- phase 2: row softmax ----------------
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
  }
  __syncthreads();

  // ---------------- phase 3 ----------------
  {
    typedef typename AtVec<TN>::type vec_t;
    const int Cz = C / zsplit;
    const int cz0 = zi * Cz;
    [[maybe_unused]] const int G = N >> 3;                    // G % D == 0 (N % 32 == 0)
    for (int cw = wave * 32 * TN; cw < Cz; cw += 4 * 32 * TN) {
      const int c0 = cz0 + cw;
      f32x16 acc[TN];
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r_ = 0; r_ < 16; ++r_) acc[t][r_] = 0.f;
      if constexpr (SPLIT) {
        const int k8 = (lane >> 5) * 8;
        const int G16 = N >> 4;                               // groups of 16 keys; G16 % 2 == 0 (N % 32 == 0)
        constexpr int DS = 2;
        const float* vcol = base + 2 * C + c0 + TN * ln + (size_t)k8 * rowstride;     // key k8, this lane's TN channels
        const float* prow = S + ln * LDS_S + k8;
        vec_t vb[DS][8];
#pragma unroll
        for (int d = 0; d < DS; ++d)
#pragma unroll
          for (int q = 0; q < 8; ++q) vb[d][q] = *reinterpret_cast<const vec_t*>(vcol + (size_t)(d * 16 + q) * rowstride);
        for (int g0 = 0; g0 < G16; g0 += DS) {
#pragma unroll
          for (int d = 0; d < DS; ++d) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(prow + (g0 + d) * 16);
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(prow + (g0 + d) * 16 + 4);
            bf16x8 pa[3];
            split3x8(p0, p1, pa[0], pa[1], pa[2]);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
              const f32x4 v0 = {at_elem<TN>(vb[d][0], t), at_elem<TN>(vb[d][1], t), at_elem<TN>(vb[d][2], t), at_elem<TN>(vb[d][3], t)};
              const f32x4 v1 = {at_elem<TN>(vb[d][4], t), at_elem<TN>(vb[d][5], t), at_elem<TN>(vb[d][6], t), at_elem<TN>(vb[d][7], t)};
              bf16x8 vv[3];
              split3x8(v0, v1, vv[0], vv[1], vv[2]);
              mfma_split6(pa, vv, acc[t]);
            }
            const size_t kn = (size_t)(min(g0 + d + DS, G16 - 1) * 16) * rowstride;
#pragma unroll
            for (int q = 0; q < 8; ++q) vb[d][q] = *reinterpret_cast<const vec_t*>(vcol + kn + (size_t)q * rowstride);
          }
        }
      }
      if constexpr (!SPLIT) {
        const float* vcol = base + 2 * C + c0 + TN * ln + (size_t)kh * rowstride;       // key kh, this lane's TN channels
        const float* prow = S + ln * LDS_S + kh;
        vec_t vb[D][4];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
          for (int q = 0; q < 4; ++q) vb[d][q] = *reinterpret_cast<const vec_t*>(vcol + (size_t)(d * 8 + q) * rowstride);
        for (int g0 = 0; g0 < G; g0 += D) {
#pragma unroll
          for (int d = 0; d < D; ++d) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(prow + (g0 + d) * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], at_elem<TN>(vb[d][q], t), acc[t], 0, 0, 0);
            const size_t kn = (size_t)(min(g0 + d + D, G - 1) * 8) * rowstride;
#pragma unroll
            for (int q = 0; q < 4; ++q) vb[d][q] = *reinterpret_cast<const vec_t*>(vcol + kn + (size_t)q * rowstride);
          }
        }
  
      }
      float* orow = out + ((size_t)b * N + m0) * C + c0 + TN * ln;
#pragma unroll
      for (int r_ = 0; r_ < 16; ++r_) {
        const int row = (r_ & 3) + 8 * (r_ >> 2) + kh;
        vec_t o;
        if constexpr (TN == 1) o = acc[0][r_];
        else {
#pragma unroll
          for (int t = 0; t < TN; ++t) o[t] = acc[t][r_];
        }
        *reinterpret_cast<vec_t*>(orow + (size_t)row * C) = o;
      }
    }
  }
}

namespace {
template <int KP, int TN, bool SPLIT>
int launch_attention_v2s(const float* qkv, int B, int N, int C, int zsplit, float* out, hipStream_t st) {
  static std::atomic<uint64_t> done{0};
  const int smem = 32 * (N + 4) * (int)sizeof(float);
  auto kern = k_attention_v2<KP, TN, SPLIT>;
  if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), 160 * 1024, done)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)((N / 32) * B * zsplit)), dim3(256), smem, st, qkv, B, N, C, zsplit, out);
  SR3_LAUNCH_CHECK("k_attention_v2");
  return SR3_OK;
}
template <int KP, int TN>
int launch_attention_v2(const float* qkv, int B, int N, int C, int zsplit, float* out, hipStream_t st, bool split) {
  return split ? launch_attention_v2s<KP, TN, true>(qkv, B, N, C, zsplit, out, st)
               : launch_attention_v2s<KP, TN, false>(qkv, B, N, C, zsplit, out, st);
}
}  // namespace

int attention_forward(const float* qkv, int B, int N, int C, float* out, hipStream_t st, int split) {
  if (C & 3) { set_error("attention: C %% 4 != 0"); return SR3_E_UNSUPPORTED; }
  if ((double)B * N * 3.0 * C >= 2147483647.0) { set_error("attention: qkv exceeds 2^31 elements"); return SR3_E_UNSUPPORTED; }
  // the staging-free kernel wherever the shape allows it (SR3_ATTN_V1=1, read once: A/B knob for the profiles)
  static const bool force_v1 = [] { const char* e = getenv("SR3_ATTN_V1"); return e && e[0] == '1'; }();
  if (!force_v1 && (N & 31) == 0 && N <= 1024 && (C % 128) == 0) {
    const int qblocks = N / 32;
    int zsplit = 1;                                       // channel split of phase 3 (each split recomputes the score strip)
    while ((long)qblocks * B * zsplit < 256 && (C / (zsplit * 2)) % 128 == 0 && zsplit < 4) zsplit *= 2;
    const int Cz = C / zsplit;
    const int tn = (Cz % 512 == 0) ? 4 : ((Cz % 256 == 0) ? 2 : 1);
    const bool pairs = N / 32 >= 8;                       // two key blocks per wave and round share the Q fragment
    if (pairs) {
      if (tn == 4) return launch_attention_v2<2, 4>(qkv, B, N, C, zsplit, out, st, split != 0);
      if (tn == 2) return launch_attention_v2<2, 2>(qkv, B, N, C, zsplit, out, st, split != 0);
      return launch_attention_v2<2, 1>(qkv, B, N, C, zsplit, out, st, split != 0);
    }
    if (tn == 4) return launch_attention_v2<1, 4>(qkv, B, N, C, zsplit, out, st, split != 0);
    if (tn == 2) return launch_attention_v2<1, 2>(qkv, B, N, C, zsplit, out, st, split != 0);
    return launch_attention_v2<1, 1>(qkv, B, N, C, zsplit, out, st, split != 0);
  }
  const int Npad = (N + 31) & ~31;
  const size_t strip = (size_t)32 * (Npad + 4);
  int nstage = 2;
  size_t smem = (strip + 2 * (size_t)AT_STAGE) * sizeof(float);
  if (smem > 160 * 1024) { nstage = 1; smem = (strip + (size_t)AT_STAGE) * sizeof(float); }
  if (smem > 160 * 1024) { set_error("attention: N=%d does not fit the LDS score strip", N); return SR3_E_UNSUPPORTED; }
  const int qblocks = (N + 31) / 32;
  const int npan = (C + 127) / 128;
  int zsplit = 1;
  while (zsplit * 2 <= npan && (long)qblocks * B * zsplit < 256) zsplit *= 2;
  static std::atomic<uint64_t> attr_done[3];
  auto kern = nstage == 2 ? k_attention<2> : k_attention<1>;
  // the attribute is an upper bound: allow the whole 160 KB LDS once per (instantiation, device)
  if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done[nstage])) return rc;
  hipLaunchKernelGGL(kern, dim3(qblocks, B, zsplit), dim3(256), smem, st, qkv, N, C, out);
  SR3_LAUNCH_CHECK("k_attention");
  return SR3_OK;
}

}  // namespace sr3
