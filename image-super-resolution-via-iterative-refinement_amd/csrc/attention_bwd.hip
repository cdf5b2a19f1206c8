// Backward of the single-head attention core (autograd of SelfAttention.forward,
// model/sr3_modules/unet.py:127-139, inside `l_pix.backward()`):
//   S = Q K^T / sqrt(C), P = softmax(S), O = P V
//   dV = P^T dO ; dP = dO V^T ; dS = P o (dP - rowsum(dP o P)) ; dQ = dS K / sqrt(C) ; dK = dS^T Q / sqrt(C)
// One workgroup owns 32 query rows of one image: it recomputes the score strip P[32][N] (as the
// forward does), builds the strip of dS / sqrt(C) next to it in LDS, writes its rows of dQ, and hands
// its contribution to dK and dV (all keys) to a slab of its own -- slab[query block][B][N][2C], plain stores, every
// element written exactly once -- which k_attn_dkv_reduce sums in query-block order: no atomics, no memset, bitwise
// reproducible (round 6; the training plan always provides the slabs).  Without slabs (the per-op ABI entry called
// with no scratch) the contributions are added with fp32 atomics as before -- dqkv is zeroed first.
// Layouts as in attention.hip: qkv / dqkv [B][N][3C] (q|k|v), dout [B][N][C].  fp32 MFMA throughout.
//
// BLOCKED = true (N too large for two full-width strips in 160 KB of LDS, e.g. the N = 1024 mid block of
// the 64 -> 512 configuration): the keys are processed in blocks of KB.  A pre-pass recomputes the scores block
// by block for the row maxima / exp-sums (online softmax), rowsum(dP o P) is taken as rowsum(dO o O) from
// the saved forward output, and the main pass handles each key block independently: its rows of dQ are
// accumulated in place across blocks (this workgroup owns them), dK / dV go out with atomics as before.
#include <algorithm>

#include "sr3_common.h"
#include "train.h"

namespace sr3 {

constexpr int AB_LDK = 36;
constexpr int AB_LDV = 132;
constexpr int AB_QK_STAGE = (32 + 128) * AB_LDK;     // floats
constexpr int AB_V_STAGE = 32 * AB_LDV;

template <int NSTAGE, bool BLOCKED>
__global__ __launch_bounds__(256) void k_attention_bwd(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                        const float* __restrict__ o_fwd, int N, int C, int KB,
                                                        float* __restrict__ dqkv, float* __restrict__ slab) {
  extern __shared__ f32x4 smem_v[];
  float* smem = reinterpret_cast<float*>(smem_v);
  const int LDS_S = (BLOCKED ? KB : ((N + 31) & ~31)) + 4;
  int kbase = 0;                          // first key of the current block
  int NK = BLOCKED ? min(KB, N) : N;      // keys in the current block
  int Npad = (NK + 31) & ~31;
  float* P = smem;                        // [32][LDS_S]  softmax probabilities
  float* D = smem + 32 * LDS_S;           // [32][LDS_S]  dP, then dS / sqrt(C)
  float* stg = smem + 64 * LDS_S;         // staging

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int m0 = blockIdx.x * 32;
  const int rs3 = 3 * C;
  const float* base = qkv + (size_t)b * N * rs3;
  const float* dob = dout + (size_t)b * N * C;
  float* dqb = dqkv + (size_t)b * N * rs3;
  const int kq = tid & 7, lrow = tid >> 3;
  const int kh = (lane >> 5) * 4;
  const float sqrt_c = sqrtf((float)C);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  constexpr int QKS = (NSTAGE == 2) ? AB_QK_STAGE : 0;

  // strip[row][key] = sum_c A[row][c] * Bm[key][c]  (A rows: a_ptr + m*a_stride, B rows: b_ptr + key*b_stride)
  auto strip_gemm = [&](const float* a_ptr, int a_stride, const float* b_ptr, int b_stride, float* strip, bool scale) {
    const int nc = (C + 31) / 32;
    const int nkb = (Npad + 127) / 128;
    const int nsteps = nkb * nc;
    f32x4 rq, rk[4];
    bool qok, kok[4];
    auto load = [&](int s) {
      const int kb = (s / nc) * 128;
      const int c = (s % nc) * 32 + kq * 4;
      const bool cv = c < C;
      const int m = m0 + lrow;
      qok = cv && m < N;
      rq = *reinterpret_cast<const f32x4*>(a_ptr + (qok ? m * a_stride + c : 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int key = kb + lrow + 32 * i;
        kok[i] = cv && key < NK;
        rk[i] = *reinterpret_cast<const f32x4*>(b_ptr + (kok[i] ? (kbase + key) * b_stride + c : 0));
      }
    };
    auto store = [&](int st) {
      float* Qs = stg + st * QKS;
      float* Ks = Qs + 32 * AB_LDK;
      *reinterpret_cast<f32x4*>(&Qs[lrow * AB_LDK + kq * 4]) = qok ? rq : zero;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(&Ks[(lrow + 32 * i) * AB_LDK + kq * 4]) = kok[i] ? rk[i] : zero;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    __syncthreads();
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
        const float* Qs = stg + cur * QKS;
        const float* Ks = Qs + 32 * AB_LDK;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(&Qs[(lane & 31) * AB_LDK + kk * 8 + kh]);
          const f32x4 k4 = *reinterpret_cast<const f32x4*>(&Ks[(wave * 32 + (lane & 31)) * AB_LDK + kk * 8 + kh]);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], k4[q], acc, 0, 0, 0);
        }
        if ((s % nc) == nc - 1) {
          const int key = kb + wave * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            strip[row * LDS_S + key] = scale ? acc[r] / sqrt_c : acc[r];
            acc[r] = 0.f;
          }
        }
      }
      if (NSTAGE == 1) __syncthreads();
      if (more) store((NSTAGE == 2) ? (cur ^ 1) : 0);
      __syncthreads();
    }
  };

  // ---- blocked path: row maxima / exp-sums over all keys (online softmax) and delta = rowsum(dO o O) ----
  float m_run = -INFINITY, l_run = 0.f, delta = 0.f;      // of row (tid >> 3); 8 threads per row
  if constexpr (BLOCKED) {
    const int row = tid >> 3, sub = tid & 7;
    for (kbase = 0; kbase < N; kbase += KB) {
      NK = min(KB, N - kbase);
      Npad = (NK + 31) & ~31;
      strip_gemm(base, rs3, base + C, rs3, P, true);
      const float* sr = P + row * LDS_S;
      float mx = m_run;
      for (int k = sub; k < NK; k += 8) mx = fmaxf(mx, sr[k]);
      mx = fmaxf(mx, __shfl_xor(mx, 1));
      mx = fmaxf(mx, __shfl_xor(mx, 2));
      mx = fmaxf(mx, __shfl_xor(mx, 4));
      float sum = 0.f;
      for (int k = sub; k < NK; k += 8) sum += expf(sr[k] - mx);
      sum += __shfl_xor(sum, 1);
      sum += __shfl_xor(sum, 2);
      sum += __shfl_xor(sum, 4);
      l_run = l_run * expf(m_run - mx) + sum;
      m_run = mx;
    }
    const int m = m0 + row;
    if (m < N) {
      for (int c = sub * 4; c < C; c += 32) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(dob + (size_t)m * C + c);
        const f32x4 o = *reinterpret_cast<const f32x4*>(o_fwd + ((size_t)b * N + m) * C + c);
        delta += g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w;
      }
    }
    delta += __shfl_xor(delta, 1);
    delta += __shfl_xor(delta, 2);
    delta += __shfl_xor(delta, 4);
  }

  const int nblocks = BLOCKED ? (N + KB - 1) / KB : 1;
  for (int blk = 0; blk < nblocks; ++blk) {
    kbase = BLOCKED ? blk * KB : 0;
    NK = BLOCKED ? min(KB, N - kbase) : N;
    Npad = (NK + 31) & ~31;
    // ---- A: P = softmax(Q K^T / sqrt(C)) (this block's keys) ----
    strip_gemm(base, rs3, base + C, rs3, P, true);
    {
      const int row = tid >> 3, sub = tid & 7;
      float* sr = P + row * LDS_S;
      if constexpr (BLOCKED) {
        for (int k = sub; k < NK; k += 8) sr[k] = expf(sr[k] - m_run) / l_run;
      } else {
        float mx = -INFINITY;
        for (int k = sub; k < NK; k += 8) mx = fmaxf(mx, sr[k]);
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        mx = fmaxf(mx, __shfl_xor(mx, 4));
        float sum = 0.f;
        for (int k = sub; k < NK; k += 8) { const float e = expf(sr[k] - mx); sr[k] = e; sum += e; }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 4);
        for (int k = sub; k < NK; k += 8) sr[k] = sr[k] / sum;
      }
      for (int k = NK + sub; k < Npad; k += 8) sr[k] = 0.f;
    }
    // ---- B: dP = dO V^T ----
    strip_gemm(dob, C, base + 2 * C, rs3, D, false);
    // ---- C: dS / sqrt(C) = P o (dP - rowsum(dP o P)) / sqrt(C)  (rows m >= N have dO = 0 => dS = 0) ----
    {
      const int row = tid >> 3, sub = tid & 7;
      const float* pr = P + row * LDS_S;
      float* dr = D + row * LDS_S;
      float rs = delta;
      if constexpr (!BLOCKED) {
        rs = 0.f;
        for (int k = sub; k < NK; k += 8) rs += dr[k] * pr[k];
        rs += __shfl_xor(rs, 1);
        rs += __shfl_xor(rs, 2);
        rs += __shfl_xor(rs, 4);
      }
      for (int k = sub; k < NK; k += 8) dr[k] = pr[k] * (dr[k] - rs) / sqrt_c;
      for (int k = NK + sub; k < Npad; k += 8) dr[k] = 0.f;
    }
    __syncthreads();

    // ---- D: dQ[32][C] (+)= (dS / sqrt(C)) K ----
    {
      const int npan = (C + 127) / 128;
      const int nk = Npad / 32;
      const int nsteps = npan * nk;
      constexpr int VS = (NSTAGE == 2) ? AB_V_STAGE : 0;
      f32x4 rv[4];
      bool vok[4];
      auto load = [&](int s) {
        const int cp = (s / nk) * 128;
        const int k0 = (s % nk) * 32;
        const int c = cp + (tid & 31) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int key = k0 + (tid >> 5) + 8 * i;
          vok[i] = key < NK && c < C;
          rv[i] = *reinterpret_cast<const f32x4*>(base + (vok[i] ? (kbase + key) * rs3 + C + c : 0));
        }
      };
      auto store = [&](int st) {
        float* Vs = stg + st * VS;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<f32x4*>(&Vs[((tid >> 5) + 8 * i) * AB_LDV + (tid & 31) * 4]) = vok[i] ? rv[i] : zero;
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
        const int cp = (s / nk) * 128;
        const int k0 = (s % nk) * 32;
        const bool wave_active = (cp + wave * 32) < C;
        if (wave_active) {
          const float* Vs = stg + cur * VS;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&D[(lane & 31) * LDS_S + k0 + kk * 8 + kh]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float bv = Vs[(kk * 8 + kh + q) * AB_LDV + wave * 32 + (lane & 31)];
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], bv, acc, 0, 0, 0);
            }
          }
          if ((s % nk) == nk - 1) {
            const int c = cp + wave * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              if (m < N && c < C) {
                float* dst = dqb + (size_t)m * rs3 + c;         // rows owned by this workgroup: plain read-add-write
                *dst = (BLOCKED && blk > 0) ? *dst + acc[r] : acc[r];
              }
              acc[r] = 0.f;
            }
          }
        }
        if (NSTAGE == 1) __syncthreads();
        if (more) store((NSTAGE == 2) ? (cur ^ 1) : 0);
        __syncthreads();
      }
    }

    // ---- E: dK += (dS/sqrt(C))^T Q ; dV += P^T dO   (this workgroup's 32 rows, this block's keys) ----
    {
      float* Qt = stg;                       // [32 rows][AB_LDV]
      float* Ot = stg + AB_V_STAGE;          // [32 rows][AB_LDV]
      const int npan = (C + 127) / 128;
      for (int pn = 0; pn < npan; ++pn) {
        const int cp = pn * 128;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (tid >> 5) + 8 * i;
          const int c = cp + (tid & 31) * 4;
          const int m = m0 + row;
          const bool ok = m < N && c < C;
          const f32x4 qv = *reinterpret_cast<const f32x4*>(base + (ok ? m * rs3 + c : 0));
          const f32x4 ov = *reinterpret_cast<const f32x4*>(dob + (ok ? m * C + c : 0));
          *reinterpret_cast<f32x4*>(&Qt[row * AB_LDV + (tid & 31) * 4]) = ok ? qv : zero;
          *reinterpret_cast<f32x4*>(&Ot[row * AB_LDV + (tid & 31) * 4]) = ok ? ov : zero;
        }
        __syncthreads();
        const int c = cp + wave * 32 + (lane & 31);
        if (cp + wave * 32 < C) {
          for (int kb = 0; kb < Npad; kb += 32) {
            f32x16 ak, av;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ak[r] = 0.f; av[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
              const int row = 2 * kk + (lane >> 5);
              const float ds = D[row * LDS_S + kb + (lane & 31)];     // A[i = key][k = row]
              const float pp = P[row * LDS_S + kb + (lane & 31)];
              const float qv = Qt[row * AB_LDV + wave * 32 + (lane & 31)];   // B[k = row][j = c]
              const float ov = Ot[row * AB_LDV + wave * 32 + (lane & 31)];
              ak = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, qv, ak, 0, 0, 0);
              av = __builtin_amdgcn_mfma_f32_32x32x2f32(pp, ov, av, 0, 0, 0);
            }
            if (c < C) {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (key < NK) {
                  if (slab) {
                    float* sl = slab + (((size_t)blockIdx.x * gridDim.y + b) * N + (kbase + key)) * (2 * C);
                    sl[c] = ak[r];
                    sl[C + c] = av[r];
                  } else {
                    atomicAdd(&dqb[(size_t)(kbase + key) * rs3 + C + c], ak[r]);
                    atomicAdd(&dqb[(size_t)(kbase + key) * rs3 + 2 * C + c], av[r]);
                  }
                }
              }
            }
          }
        }
      }
    }
  }
}

// dK / dV = the query blocks' slabs summed in block order (fixed order: bitwise reproducible); one thread per (b, key, channel quad of 2C)
__global__ __launch_bounds__(256) void k_attn_dkv_reduce(const float* __restrict__ slab, int nqb, int B, int N, int C, float* __restrict__ dqkv) {
  const size_t quads = (size_t)B * N * (2 * C / 4);
  const size_t per = (size_t)B * N * 2 * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / (2 * C / 4);
    const int c = (int)(i - row * (2 * C / 4)) * 4;
    f32x4 a = *reinterpret_cast<const f32x4*>(slab + i * 4);
    for (int q = 1; q < nqb; ++q) a += *reinterpret_cast<const f32x4*>(slab + (size_t)q * per + i * 4);
    *reinterpret_cast<f32x4*>(dqkv + row * (3 * C) + C + c) = a;
  }
}

size_t attention_backward_scratch_bytes(int B, int N, int C) { return (size_t)((N + 31) / 32) * B * N * 2 * C * sizeof(float); }

int attention_backward(const float* qkv, const float* dout, const float* out_fwd, int B, int N, int C, float* dqkv, hipStream_t st,
                       float* scratch, size_t scratch_bytes) {
  if (C & 3) { set_error("attention_bwd: C %% 4 != 0"); return SR3_E_UNSUPPORTED; }
  if ((double)B * N * 3.0 * C >= 2147483647.0) { set_error("attention_bwd: qkv exceeds 2^31 elements"); return SR3_E_UNSUPPORTED; }
  const size_t lds_max = 160 * 1024;
  const size_t stage2 = 2 * (size_t)(AB_QK_STAGE > AB_V_STAGE ? AB_QK_STAGE : AB_V_STAGE);
  const size_t stage1 = (size_t)(AB_QK_STAGE > 2 * AB_V_STAGE ? AB_QK_STAGE : 2 * AB_V_STAGE);
  const int Npad = (N + 31) & ~31;
  auto bytes = [](int width, size_t stage_f) { return ((size_t)64 * (width + 4) + stage_f) * sizeof(float); };
  int nstage = 2, KB = 0;
  size_t smem = bytes(Npad, stage2);
  if (smem > lds_max) { nstage = 1; smem = bytes(Npad, stage1); }
  if (smem > lds_max) {
    // key-blocked path: the widest block (a multiple of 128 keys) whose two strips fit next to double-buffered staging
    if (!out_fwd) { set_error("attention_bwd: N=%d needs the forward output for the key-blocked path", N); return SR3_E_BADARG; }
    nstage = 2;
    KB = 128;
    while (bytes(KB + 128, stage2) <= lds_max) KB += 128;
    smem = bytes(KB, stage2);
  }
  const bool slabs = scratch != nullptr && scratch_bytes >= attention_backward_scratch_bytes(B, N, C);
  if (scratch && !slabs) { set_error("attention_bwd: scratch too small for the dK / dV slabs (%zu < %zu)", scratch_bytes, attention_backward_scratch_bytes(B, N, C)); return SR3_E_NOMEM; }
  if (!slabs) SR3_HIP(hipMemsetAsync(dqkv, 0, (size_t)B * N * 3 * C * sizeof(float), st));        // (atomics path only)
  static std::atomic<uint64_t> attr_done[3];
  const int which = KB ? 2 : nstage - 1;
  auto kern = KB ? k_attention_bwd<2, true> : (nstage == 2 ? k_attention_bwd<2, false> : k_attention_bwd<1, false>);
  // the attribute is an upper bound: allow the whole LDS once per (instantiation, device)
  if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), (int)lds_max, attr_done[which])) return rc;
  hipLaunchKernelGGL(kern, dim3((N + 31) / 32, B), dim3(256), smem, st, qkv, dout, out_fwd, N, C, KB, dqkv, slabs ? scratch : nullptr);
  SR3_LAUNCH_CHECK("k_attention_bwd");
  if (slabs) {
    const size_t quads = (size_t)B * N * (2 * C / 4);
    int blocks = (int)std::min<size_t>((quads + 255) / 256, 8192);
    hipLaunchKernelGGL(k_attn_dkv_reduce, dim3(blocks), dim3(256), 0, st, scratch, (N + 31) / 32, B, N, C, dqkv);
    SR3_LAUNCH_CHECK("k_attn_dkv_reduce");
  }
  return SR3_OK;
}

}  // namespace sr3

extern "C" int sr3_attention_bwd_f32(const float* qkv, const float* dout, const float* out_fwd, int B, int N, int C, float* dqkv,
                                     void* stream) {
  if (!qkv || !dout || !dqkv) { sr3::set_error("null argument"); return SR3_E_BADARG; }
  return sr3::attention_backward(qkv, dout, out_fwd, B, N, C, dqkv, static_cast<hipStream_t>(stream), nullptr, 0);
}
extern "C" size_t sr3_attention_bwd_scratch_bytes(int B, int N, int C) { return sr3::attention_backward_scratch_bytes(B, N, C); }
extern "C" int sr3_attention_bwd_ex_f32(const float* qkv, const float* dout, const float* out_fwd, int B, int N, int C, float* dqkv,
                                        void* scratch, size_t scratch_bytes, void* stream) {
  if (!qkv || !dout || !dqkv || !scratch) { sr3::set_error("null argument"); return SR3_E_BADARG; }
  return sr3::attention_backward(qkv, dout, out_fwd, B, N, C, dqkv, static_cast<hipStream_t>(stream), static_cast<float*>(scratch), scratch_bytes);
}
