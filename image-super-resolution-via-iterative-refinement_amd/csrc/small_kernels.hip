// HBM-bound helpers of the SR3 / DDPM hot path (gfx950): GroupNorm statistics and folding,
// the 6->C input conv (NCHW -> NHWC), the C->3 output Block (NHWC -> NCHW), the noise-level /
// timestep embedding with every FiLM projection, and the fused reverse-step / q_sample updates.
#include <string.h>

#include "sr3_common.h"

namespace sr3 {

__device__ __forceinline__ float silu_s(float v) { return SR3_SILU(v); }
// separately rounded product / sum: the empty asm makes the value opaque so hipcc (default
// -ffp-contract=fast) cannot fuse it into an fma -- bit parity with torch's elementwise ops.
__device__ __forceinline__ float mul_rn(float a, float b) { float r = a * b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ float add_rn(float a, float b) { float r = a + b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ float sub_rn(float a, float b) { float r = a - b; asm volatile("" : "+v"(r)); return r; }

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics (nn.GroupNorm, model/sr3_modules/unet.py:84,119) as PARTIAL per-(image,
// channel) {sum, sumsq} in double:  stat[b][t][c][2], t < T partials per image.  Producers write
// disjoint partials with plain stores (this kernel: one per pixel slice; the halo conv: one per
// spatial tile) and the fold kernel sums them in a fixed order -- no atomics, no memset, bitwise
// reproducible.  Keeping *channel* sums lets one pass serve any grouping, including groups that
// straddle a skip-concat seam (unet.py:255).
// x: NHWC [B][HW][C].  grid (T slices, cblocks, B); LQ lanes across channel quads, 256/LQ across pixels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chan_stats(const float* __restrict__ x, int HW, int C, int LQ,
                                                     int pix_per_block, double* __restrict__ stat) {
  __shared__ double red[256 * 8];
  const int tid = threadIdx.x;
  const int nq = C >> 2;
  const int ql = tid % LQ, pl = tid / LQ, PP = 256 / LQ;
  const int q = blockIdx.y * LQ + ql;
  const int b = blockIdx.z;
  const int T = gridDim.x;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(HW, p0 + pix_per_block);
  double s[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (q < nq) {
    const float* base = x + (size_t)b * HW * C + q * 4;
    for (int p = p0 + pl; p < p1; p += PP) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * C);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const double d = (double)v[e]; s[e] += d; s2[e] += d * d; }
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
    double* o = stat + (((size_t)b * T + blockIdx.x) * C + q * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = s[e]; o[2 * e + 1] = s2[e]; }
  }
}

static void chan_stats_geometry(int B, int HW, int C, int* LQ_, int* cblocks_, int* ppb_, int* slices_) {
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

int chan_stats_slices(int B, int HW, int C) {
  int LQ, cb, ppb, sl;
  chan_stats_geometry(B, HW, C, &LQ, &cb, &ppb, &sl);
  return sl;
}

int chan_stats(const float* x, int B, int HW, int C, double* stat, hipStream_t st) {
  if (C & 3) { set_error("chan_stats: C %% 4 != 0 (%d)", C); return SR3_E_UNSUPPORTED; }
  int LQ, cblocks, ppb, slices;
  chan_stats_geometry(B, HW, C, &LQ, &cblocks, &ppb, &slices);
  hipLaunchKernelGGL(k_chan_stats, dim3(slices, cblocks, B), dim3(256), 0, st, x, HW, C, LQ, ppb, stat);
  SR3_LAUNCH_CHECK("k_chan_stats");
  return SR3_OK;
}

// ---------------------------------------------------------------------------------------------
// GroupNorm fold: one wave per (image, group) sums the partials of the group's channels over the
// virtual concat [src0 | src1] in a fixed order, then writes scale = rstd * gamma and
// shift = beta - mean * scale per channel.  Biased variance, eps inside the sqrt (torch GroupNorm).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_gn_finalize(const double* __restrict__ st0, int C0, int T0,
                                                     const double* __restrict__ st1, int C1, int T1, int HW,
                                                     int groups, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps,
                                                     float* __restrict__ ss, float* __restrict__ mr) {
  const int C = C0 + C1;
  const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
  const int cpg = C / groups;
  const int lane = threadIdx.x;
  double s = 0.0, s2 = 0.0;
  // the group's channels [c_lo, c_hi) split at the concat seam; each part is a flat (channel, partial)
  // index space walked by the 64 lanes: independent loads, fixed summation order
  const int c_lo = g * cpg, c_hi = c_lo + cpg;
  const int n0 = max(0, min(c_hi, C0) - c_lo);            // channels of this group that live in src0
  // four partials per lane in flight (round 6): the partials were written by the previous kernel on other XCDs, every load is a
  // trip to memory, and one load per loop iteration made the wave wait for each in turn (5.7 us per fold, 61 folds per step).
  // The additions keep the order of the one-at-a-time loop: the sums are bit-identical.
  typedef double d2 __attribute__((ext_vector_type(2)));
  auto walk = [&](const double* st, int Cs, int T, int c_first, int n) {
    const int total = n * T;
    auto at = [&](int idx) {
      const int k = idx / T, t = idx - k * T;
      return *reinterpret_cast<const d2*>(st + (((size_t)b * T + t) * Cs + (c_first + k)) * 2);
    };
    int idx = lane;
    for (; idx + 192 < total; idx += 256) {
      const d2 q0 = at(idx), q1 = at(idx + 64), q2 = at(idx + 128), q3 = at(idx + 192);
      s += q0[0]; s2 += q0[1]; s += q1[0]; s2 += q1[1]; s += q2[0]; s2 += q2[1]; s += q3[0]; s2 += q3[1];
    }
    for (; idx < total; idx += 64) { const d2 q = at(idx); s += q[0]; s2 += q[1]; }
  };
  walk(st0, C0, T0, c_lo, n0);
  const int n1 = cpg - n0;
  const int c1_lo = max(c_lo, C0) - C0;
  if (n1 > 0) walk(st1, C1, T1, c1_lo, n1);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { s += __shfl_xor(s, m); s2 += __shfl_xor(s2, m); }
  const double cnt = (double)HW * cpg;
  const double mean = s / cnt;
  double var = s2 / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (mr && lane == 0) { mr[((size_t)b * groups + g) * 2] = (float)mean; mr[((size_t)b * groups + g) * 2 + 1] = rstd; }
  for (int k = lane; k < cpg; k += 64) {
    const int c = g * cpg + k;
    const float sc = rstd * gamma[c];
    float* o = ss + ((size_t)b * C + c) * 2;
    o[0] = sc;
    o[1] = beta[c] - (float)mean * sc;
  }
}

int gn_finalize(const double* stat0, int C0, int T0, const double* stat1, int C1, int T1, int B, int HW, int groups,
                const float* gamma, const float* beta, float eps, float* ss, hipStream_t st, float* mr) {
  const int C = C0 + C1;
  if (groups <= 0 || C % groups) { set_error("gn_finalize: C=%d not divisible by groups=%d", C, groups); return SR3_E_BADARG; }
  hipLaunchKernelGGL(k_gn_finalize, dim3(B * groups), dim3(64), 0, st, stat0, C0, T0, stat1, C1, T1, HW, groups,
                     gamma, beta, eps, ss, mr);
  SR3_LAUNCH_CHECK("k_gn_finalize");
  return SR3_OK;
}

// ---------------------------------------------------------------------------------------------
// Rows + fold (FoldTail, sr3_common.h; round 6): one workgroup per (image, group of the CONSUMER's GroupNorm).  It walks all pixels of
// the image over the group's channels that live in THIS tensor -- REDUCE: summing the split-K slabs of the conv that produced it, with
// that conv's epilogue (bias, FiLM, residual), i.e. it replaces k_splitk_reduce; otherwise reading the finished tensor, i.e. it replaces
// k_chan_stats -- keeps {sum, sumsq} in double per channel (lanes along channel quads, a fixed-order LDS fold over the row lanes), writes
// them as the tensor's partials (T = 1), adds the group's channels of the OTHER concat source from its partials, and writes the
// consumer's (scale, shift) pairs: what k_gn_finalize would have done in a launch of its own.
// ---------------------------------------------------------------------------------------------
template <bool REDUCE>
__global__ __launch_bounds__(256) void k_rows_fold(const ConvParams p, const FoldTail f, const float* __restrict__ x, int C, int HW,
                                                    double* __restrict__ stat) {
  __shared__ double red[256 * 8];
  __shared__ double chs[2 * 128];       // per-channel {sum, sumsq} of the group, concat order
  __shared__ float mrs[2];
  const int tid = threadIdx.x;
  // XCD-aware order (last session of round 6): workgroups are dealt to the 8 XCDs round-robin, and neighbouring groups of an image share
  // 128-byte lines of every row (a group of 16 channels is half a line) -- consecutive (image, group) pairs go to ONE XCD, behind one L2
#ifdef SR3_FOLD_NO_XCD
  const int lin = blockIdx.x;
#else
  int lin;
  {
    const int nwg = gridDim.x, w = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = w & 7;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
  }
#endif
  const int b = lin / f.groups, g = lin - b * f.groups;
  const int cpg = f.Ctot / f.groups;
  const int glo = g * cpg, ghi = glo + cpg;
  const int a_lo = max(glo, f.c_off) - f.c_off, a_hi = min(ghi, f.c_off + C) - f.c_off;      // this tensor's channels of the group
  const int na = a_hi > a_lo ? a_hi - a_lo : 0;
  const int nq = na >> 2;
  int nqp = 1;
  while (nqp < nq) nqp <<= 1;
  const int R = 256 / nqp, tq = tid % nqp, tr = tid / nqp;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (tq < nq) {
    const int n = a_lo + 4 * tq;
    const size_t M = (size_t)p.B * HW;
    f32x4 cb = {0.f, 0.f, 0.f, 0.f};
    if constexpr (REDUCE) {
      if (p.bias) cb += *reinterpret_cast<const f32x4*>(p.bias + n);
      if (p.x2_w && p.x2_bias) cb += *reinterpret_cast<const f32x4*>(p.x2_bias + n);
    }
    for (int r = tr; r < HW; r += R) {
      const size_t m = (size_t)b * HW + r;
      f32x4 v;
      if constexpr (REDUCE) {
        // the order of k_splitk_reduce: bias first, the slabs in split order, then FiLM and the residual -- bit-identical outputs
        v = cb;
        auto slab = [&](int s) { return *reinterpret_cast<const f32x4*>(p.partial + ((size_t)s * M + m) * p.Cout + n); };
        int s = 0;
        for (; s + 4 <= p.ksplit; s += 4) {
          const f32x4 q0 = slab(s), q1 = slab(s + 1), q2 = slab(s + 2), q3 = slab(s + 3);
          v += q0; v += q1; v += q2; v += q3;
        }
        for (; s < p.ksplit; ++s) v += slab(s);
        if (p.film) v += *reinterpret_cast<const f32x4*>(p.film + (size_t)b * p.film_stride + n);
        if (p.res0) {
          if (n < p.RC0) v += *reinterpret_cast<const f32x4*>(p.res0 + m * p.RC0 + n);
          else v += *reinterpret_cast<const f32x4*>(p.res1 + m * p.RC1 + (n - p.RC0));
        }
        *reinterpret_cast<f32x4*>(p.out + m * p.Cout + n) = v;
      } else {
        v = *reinterpret_cast<const f32x4*>(x + m * C + n);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { const double dv = (double)v[e]; s1[e] += dv; s2[e] += dv * dv; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = s1[e]; red[tid * 8 + 4 + e] = s2[e]; }
  __syncthreads();
  // this tensor's channels: fixed-order sum over the row lanes, one thread per channel
  if (tid < na) {
    const int q = tid >> 2, e = tid & 3;
    double a1 = 0.0, a2 = 0.0;
    for (int k = 0; k < R; ++k) { a1 += red[(k * nqp + q) * 8 + e]; a2 += red[(k * nqp + q) * 8 + 4 + e]; }
    double* o = stat + ((size_t)b * C + a_lo + tid) * 2;
    o[0] = a1; o[1] = a2;
    const int cc = f.c_off + a_lo + tid - glo;            // position inside the group (concat order)
    chs[2 * cc] = a1; chs[2 * cc + 1] = a2;
  }
  // the other source's channels of the group: its partials in slice order, one thread per channel
  if (f.ostat) {
    const int o_lo = max(glo, f.o_off) - f.o_off, o_hi = min(ghi, f.o_off + f.oC) - f.o_off;
    const int no = o_hi > o_lo ? o_hi - o_lo : 0;
    const int t2 = tid - 128;             // (the upper half of the block: runs beside the loop above)
    if (t2 >= 0 && t2 < no) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      double a1 = 0.0, a2 = 0.0;
      const double* q = f.ostat + ((size_t)b * f.oT * f.oC + o_lo + t2) * 2;
      int t = 0;
      for (; t + 4 <= f.oT; t += 4) {
        const d2 q0 = *reinterpret_cast<const d2*>(q + (size_t)t * f.oC * 2), q1 = *reinterpret_cast<const d2*>(q + (size_t)(t + 1) * f.oC * 2),
                 q2 = *reinterpret_cast<const d2*>(q + (size_t)(t + 2) * f.oC * 2), q3 = *reinterpret_cast<const d2*>(q + (size_t)(t + 3) * f.oC * 2);
        a1 += q0[0]; a2 += q0[1]; a1 += q1[0]; a2 += q1[1]; a1 += q2[0]; a2 += q2[1]; a1 += q3[0]; a2 += q3[1];
      }
      for (; t < f.oT; ++t) { const d2 q0 = *reinterpret_cast<const d2*>(q + (size_t)t * f.oC * 2); a1 += q0[0]; a2 += q0[1]; }
      const int cc = f.o_off + o_lo + t2 - glo;
      chs[2 * cc] = a1; chs[2 * cc + 1] = a2;
    }
  }
  __syncthreads();
  if (tid == 0) {
    double s = 0.0, sq = 0.0;
    for (int k = 0; k < cpg; ++k) { s += chs[2 * k]; sq += chs[2 * k + 1]; }
    const double cnt = (double)HW * cpg;
    const double mean = s / cnt;
    double var = sq / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
    mrs[0] = (float)mean; mrs[1] = rstd;
    if (f.mr) { f.mr[((size_t)b * f.groups + g) * 2] = (float)mean; f.mr[((size_t)b * f.groups + g) * 2 + 1] = rstd; }
  }
  __syncthreads();
  if (tid < cpg) {
    const int c = glo + tid;
    const float sc = mrs[1] * f.gamma[c];
    float* o = f.ss + ((size_t)b * f.Ctot + c) * 2;
    o[0] = sc;
    o[1] = f.beta[c] - mrs[0] * sc;
  }
}

// the group grid covers this tensor in whole channel quads, a group has at most 128 channels (LDS table) and at most 64 of them per
// source side (lanes: 16 quads here, 128 threads for the other source)
bool fold_tail_fits(int C, int Ctot, int c_off, int groups) {
  if (groups <= 0 || Ctot % groups) return false;
  const int cpg = Ctot / groups;
  return (cpg & 3) == 0 && cpg <= 64 && (c_off & 3) == 0 && (C & 3) == 0;
}

int chan_stats_fold(const float* x, int B, int HW, int C, double* stat, const FoldTail& f, hipStream_t st) {
  if (!fold_tail_fits(C, f.Ctot, f.c_off, f.groups)) { set_error("chan_stats_fold: grouping does not fit"); return SR3_E_UNSUPPORTED; }
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.B = B;
  hipLaunchKernelGGL(k_rows_fold<false>, dim3(B * f.groups), dim3(256), 0, st, p, f, x, C, HW, stat);
  SR3_LAUNCH_CHECK("k_rows_fold");
  return SR3_OK;
}

int splitk_reduce_fold(const ConvParams& p, const FoldTail& f, hipStream_t st) {
  if (!fold_tail_fits(p.Cout, f.Ctot, f.c_off, f.groups) || !p.ostat) { set_error("splitk_reduce_fold: grouping does not fit"); return SR3_E_UNSUPPORTED; }
  hipLaunchKernelGGL(k_rows_fold<true>, dim3(p.B * f.groups), dim3(256), 0, st, p, f, static_cast<const float*>(nullptr), p.Cout,
                     p.Ho * p.Wo, p.ostat);
  SR3_LAUNCH_CHECK("k_rows_fold");
  return SR3_OK;
}

// ---------------------------------------------------------------------------------------------
// Self-test of split3_pair (sr3_common.h): every split kernel rests on x == h + m + l holding EXACTLY, and the shipped form of
// the helper depends on two things no compiler promises -- the bf16 selector pairs staying in registers (folded into the inline
// constant -1.0 they read as (0, -1) for both elements) and v_dot2c_f32_bf16 producing the exactly representable residual.  One
// workgroup splits a set of fp32 patterns (random bits over the normal range, powers of two, values one ulp either side of a bf16
// rounding boundary, zeros, a value with 24 set mantissa bits) and counts the elements whose three terms do not add back to the
// input bit for bit, or whose terms are not bf16-exact halves of the residual chain.  Run by __graft_entry__.smoke() and a GPU test.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_split3_selftest(int n, unsigned seed, int* __restrict__ bad) {
  int nbad = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float x[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned r = hash32((unsigned)(2 * i + e) * 0x9E3779B9u + seed);
      unsigned bits;
      switch (i & 7) {
        case 0: bits = (r & 0x807fffffu) | (((r >> 23) % 200u + 27u) << 23); break;          // random mantissa, exponent 27..226
        case 1: bits = ((r >> 23) % 200u + 27u) << 23; break;                                 // powers of two
        case 2: bits = (r & 0xffff0000u & 0x807fffffu) | (100u << 23) | 0x00007fffu; break;   // just below a bf16 tie
        case 3: bits = (r & 0xffff0000u & 0x807fffffu) | (140u << 23) | 0x00008001u; break;   // just above a bf16 tie
        case 4: bits = (r & 0x80000000u) | (127u << 23) | 0x007fffffu; break;                 // 24 set significand bits
        case 5: bits = r & 0x80000000u; break;                                                // +-0
        case 6: bits = (r & 0x807fffffu) | (127u << 23); break;                               // [1, 2)
        default: bits = (r & 0xffff8000u & 0x807fffffu) | (90u << 23) | 0x00008000u; break;   // exact ties
      }
      x[e] = __builtin_bit_cast(float, bits);
    }
    bf16x2 h, m, l;
    split3_pair(x[0], x[1], h, m, l);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float back = ((float)h[e] + (float)m[e]) + (float)l[e];          // every partial sum is exactly representable
      if (__builtin_bit_cast(unsigned, back) != __builtin_bit_cast(unsigned, x[e]) && !(back == 0.f && x[e] == 0.f)) ++nbad;
    }
  }
  if (nbad) atomicAdd(bad, nbad);
}

int split3_selftest(int* bad_dev, hipStream_t st) {
  SR3_HIP(hipMemsetAsync(bad_dev, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_split3_selftest, dim3(64), dim3(256), 0, st, 1 << 20, 0x5eedu, bad_dev);
  SR3_LAUNCH_CHECK("k_split3_selftest");
  return SR3_OK;
}

// ---------------------------------------------------------------------------------------------
// Input conv (unet.py:193-194): 3x3 pad 1 over the virtual concat of two NCHW tensors
// (`torch.cat([condition_x, x], dim=1)`, diffusion.py:157) -> NHWC.  K = 9*(Ca+Cb) is tiny (54),
// so this is a direct fp32 FMA kernel bound by the NHWC write: lanes = 64 consecutive pixels
// (coalesced NCHW reads), the 4 waves of a block each own 16 of every 64 output channels and read
// their weights as LDS broadcasts.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_in_nchw(const float* __restrict__ a, int Ca,
                                                       const float* __restrict__ bsrc, int Cb, int B, int H,
                                                       int W, const float* __restrict__ w,
                                                       const float* __restrict__ bias, int Cout,
                                                       float* __restrict__ out) {
  extern __shared__ f32x4 smem_v[];
  float* wl = reinterpret_cast<float*>(smem_v);          // [K][CoutP]  (k-major, CoutP = Cout rounded to 64)
  const int Cin = Ca + Cb;
  const int K = 9 * Cin;
  const int CoutP = (Cout + 63) & ~63;
  const int tid = threadIdx.x;
  for (int i = tid; i < K * CoutP; i += 256) {
    const int k = i / CoutP, n = i - k * CoutP;
    const int tap = k / Cin, c = k - tap * Cin;
    wl[i] = (n < Cout) ? w[((size_t)n * 9 + tap) * Cin + c] : 0.f;
  }
  __syncthreads();
  const int HWp = H * W;
  const long M = (long)B * HWp;
  const long m = (long)blockIdx.x * 64 + (tid & 63);
  const int cog = tid >> 6;                                // wave index = output-channel group
  if (m >= M) return;
  const int b = (int)(m / HWp);
  const int rem = (int)(m - (long)b * HWp);
  const int oh = rem / W, ow = rem - oh * W;
  for (int cobase = 0; cobase < Cout; cobase += 64) {
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
      const int ih = oh + tap / 3 - 1, iw = ow + tap % 3 - 1;
      const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      for (int c = 0; c < Cin; ++c) {
        float v = 0.f;
        if (ok) {
          v = (c < Ca) ? a[((size_t)b * Ca + c) * HWp + (size_t)ih * W + iw]
                       : bsrc[((size_t)b * Cb + (c - Ca)) * HWp + (size_t)ih * W + iw];
        }
        const float* wr = wl + (size_t)(tap * Cin + c) * CoutP + cobase + cog * 16;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const f32x4 ww = *reinterpret_cast<const f32x4*>(wr + e4 * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e4 * 4 + e] = fmaf(v, ww[e], acc[e4 * 4 + e]);
        }
      }
    }
    const int co0 = cobase + cog * 16;
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const int co = co0 + e4 * 4;
      if (co < Cout) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[e4 * 4 + e] + (bias ? bias[co + e] : 0.f);
        *reinterpret_cast<f32x4*>(out + (size_t)m * Cout + co) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same conv on the fp32 MFMA (round 3; the VALU form above stays for shapes this one does not take).  GEMM view:
// out[m][n] = sum_k x[m][k] w[n][k], k = tap * CIN + c (the OHWI filter row as it lies in memory), two k per
// v_mfma_f32_32x32x2_f32.  A wave owns 32 consecutive pixels of one image row: lane (pixel l & 31, k-half l >> 5) reads its A
// value straight from the NCHW plane (32 lanes = 128 contiguous bytes), no LDS anywhere; the whole filter bank (KS x NB
// registers) is loaded once per wave and reused for `bpw` pixel blocks.  Epilogue: + bias, NHWC stores (a register's 32
// lanes = 32 consecutive channels of a pixel), and the per-(image, slice, channel) GroupNorm partial sums in double --
// the stand-alone statistics pass over this 67 MB tensor goes away (slice = the bpw blocks of one wave).
// Needs W % 32 == 0, Cout = 32 NB, CIN in {3, 6}.
// ---------------------------------------------------------------------------------------------
template <int CIN, int NB>
__global__ __launch_bounds__(256) void k_conv_in_mfma(const float* __restrict__ a, int Ca, const float* __restrict__ bsrc,
                                                       int Cb, int B, int H, int W, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       double* __restrict__ ostat, int bpw, int T) {
  constexpr int K = 9 * CIN, KS = (K + 1) / 2, Cout = 32 * NB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ln = lane & 31, h = lane >> 5;
  const int HW = H * W, bpi = HW >> 5;                       // 32-pixel blocks per image
  const int task = blockIdx.x * 4 + wave;
  if (task >= B * (bpi / bpw)) return;                       // (no barrier in this kernel)
  const int blk0 = task * bpw;
  const int b = blk0 / bpi;
  const int slice = (blk0 - b * bpi) / bpw;

  float bw[KS][NB], bn[NB];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int k = 2 * s + h;
      bw[s][nb] = k < K ? w[(size_t)(nb * 32 + ln) * K + k] : 0.f;
    }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bn[nb] = bias ? bias[nb * 32 + ln] : 0.f;

  const float* plane[CIN];                                   // channel planes of image b: the virtual concat [a | bsrc]
#pragma unroll
  for (int c = 0; c < CIN; ++c)
    plane[c] = c < Ca ? a + ((size_t)b * Ca + c) * HW : bsrc + ((size_t)b * Cb + (c - Ca)) * HW;

  auto load_block = [&](int blk, float (&av)[KS]) {
    const int p0 = (blk - b * bpi) * 32;                     // first pixel of the block inside the image (one image row)
    const int oh = p0 / W, ow = p0 - oh * W + ln;
    const int pixoff = oh * W + ow;
    const bool up = oh > 0, down = oh < H - 1, left = ow > 0, right = ow < W - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      constexpr int dummy = 0; (void)dummy;
      const int k0 = 2 * s, k1 = 2 * s + 1;                  // compile-time after unrolling
      const int t0 = k0 / CIN, c0 = k0 - t0 * CIN, t1 = k1 / CIN, c1 = (k1 < K) ? k1 - t1 * CIN : 0;
      const int dy0 = t0 / 3 - 1, dx0 = t0 % 3 - 1, dy1 = (k1 < K) ? t1 / 3 - 1 : 0, dx1 = (k1 < K) ? t1 % 3 - 1 : 0;
      const bool v0 = (dy0 < 0 ? up : (dy0 > 0 ? down : true)) && (dx0 < 0 ? left : (dx0 > 0 ? right : true));
      const bool v1 = (k1 < K) && (dy1 < 0 ? up : (dy1 > 0 ? down : true)) && (dx1 < 0 ? left : (dx1 > 0 ? right : true));
      const bool valid = h ? v1 : v0;
      const float* pl = h ? plane[c1] : plane[c0];
      const int d = h ? dy1 * W + dx1 : dy0 * W + dx0;
      const float v = pl[pixoff + (valid ? d : 0)];          // unconditional load (the centre pixel when the tap is padding)
      av[s] = valid ? v : 0.f;
    }
  };

  double s1[NB], s2[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) { s1[nb] = 0.0; s2[nb] = 0.0; }
  float av[KS], an[KS];
  load_block(blk0, av);
  for (int i = 0; i < bpw; ++i) {
    load_block(blk0 + min(i + 1, bpw - 1), an);              // (the last block re-fetches itself)
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bw[s][nb], acc[nb], 0, 0, 0);
    const size_t m0 = (size_t)(blk0 + i) * 32;               // NHWC row of the block's first pixel
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = acc[nb][r] + bn[nb];
        out[(m0 + row) * Cout + nb * 32 + ln] = v;
        if (ostat) { const double dv = (double)v; s1[nb] += dv; s2[nb] += dv * dv; }
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) av[s] = an[s];
  }
  if (ostat) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const double t1 = s1[nb] + __shfl_xor(s1[nb], 32), t2 = s2[nb] + __shfl_xor(s2[nb], 32);
      if (h == 0) {
        double* o = ostat + (((size_t)b * T + slice) * Cout + nb * 32 + ln) * 2;
        o[0] = t1; o[1] = t2;
      }
    }
  }
}

// 32-pixel blocks per wave of the MFMA form (0: the shape stays on the VALU kernel); also the statistics slices per image
static int conv_in_bpw(int Cin, int H, int W, int Cout) {
  if ((Cin != 3 && Cin != 6) || (W & 31) || (Cout & 31) || Cout > 128 || Cout < 32) return 0;
  const int bpi = H * W / 32;
  return (bpi % 4 == 0) ? 4 : 1;
}
int conv_in_stat_slices(int Cin, int H, int W, int Cout) {
  const int bpw = conv_in_bpw(Cin, H, W, Cout);
  return bpw ? H * W / 32 / bpw : 0;
}

int conv_in_nchw(const float* a, int Ca, const float* b, int Cb, int B, int H, int W, const float* w,
                 const float* bias, int Cout, float* out, double* ostat, hipStream_t st) {
  if (const int bpw = conv_in_bpw(Ca + Cb, H, W, Cout)) {
    const int T = H * W / 32 / bpw;
    const long tasks = (long)B * T;
    const dim3 grid((unsigned)((tasks + 3) / 4));
#define SR3_CI(CIN, NB) hipLaunchKernelGGL((k_conv_in_mfma<CIN, NB>), grid, dim3(256), 0, st, a, Ca, b, Cb, B, H, W, w, bias, out, ostat, bpw, T)
    const int nb = Cout / 32;
    if (Ca + Cb == 6) { if (nb == 1) SR3_CI(6, 1); else if (nb == 2) SR3_CI(6, 2); else if (nb == 3) SR3_CI(6, 3); else SR3_CI(6, 4); }
    else { if (nb == 1) SR3_CI(3, 1); else if (nb == 2) SR3_CI(3, 2); else if (nb == 3) SR3_CI(3, 3); else SR3_CI(3, 4); }
#undef SR3_CI
    SR3_LAUNCH_CHECK("k_conv_in_mfma");
    return SR3_OK;
  }
  if (ostat) { set_error("conv_in: fused output statistics need the MFMA form (W %% 32 == 0, Cout %% 32 == 0, 3 or 6 input channels)"); return SR3_E_UNSUPPORTED; }
  if (Cout & 3) { set_error("conv_in: Cout %% 4 != 0"); return SR3_E_UNSUPPORTED; }
  const int CoutP = (Cout + 63) & ~63;
  const size_t smem = (size_t)9 * (Ca + Cb) * CoutP * sizeof(float);
  if (smem > 64 * 1024) { set_error("conv_in: weights do not fit LDS (%zu B)", smem); return SR3_E_UNSUPPORTED; }
  const long M = (long)B * H * W;
  hipLaunchKernelGGL(k_conv_in_nchw, dim3((unsigned)((M + 63) / 64)), dim3(256), smem, st, a, Ca, b, Cb, B, H, W, w,
                     bias, Cout, out);
  SR3_LAUNCH_CHECK("k_conv_in_nchw");
  return SR3_OK;
}

// ---------------------------------------------------------------------------------------------
// Output Block (unet.py:233,259): GroupNorm -> Swish -> conv3x3 C -> Cout (3).  N = 3 is no GEMM;
// the kernel is bound by one read of the NHWC input.  A block owns an 8x32 pixel tile; per 16-
// channel chunk it stages the (10x34) halo tile *after* GN+SiLU in LDS (each element transformed
// once, zero padding applied after the activation as the reference does), then every thread
// accumulates its pixel's Cout outputs from LDS.  Writes NCHW (the public layout of eps).
// ---------------------------------------------------------------------------------------------
constexpr int OT_H = 8, OT_W = 32, OT_CK = 16, OT_LD = 20;
// COUT: output channels (1..4); FULL: C is a multiple of the 16-channel chunk (no per-quad bound checks)
// FUSE (sr3_reverse_step): the reverse-step update of the image this eps belongs to in the epilogue -- the element a thread produces IS the
// element of eps the elementwise update (sr3 diffusion.py:141-149,162-174) needs, in the same NCHW position -- with k_p_sample_update's
// separately rounded operations (bit-identical to the two-kernel form), and the loop counter's decrement by one thread
template <int COUT, bool FULL, bool FUSE>
__global__ __launch_bounds__(256) void k_conv_out_nchw(const float* __restrict__ x, const float* __restrict__ ss,
                                                        int B, int H, int W, int C, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int Cout,
                                                        float* __restrict__ out, const StepFuse f) {
  __shared__ f32x4 tile_v[(OT_H + 2) * (OT_W + 2) * OT_LD / 4];
  float* tile = reinterpret_cast<float*>(tile_v);
  const int tid = threadIdx.x;
  const int tx = tid & 31, ty = tid >> 5;
  const int tiles_w = (W + OT_W - 1) / OT_W, tiles_h = (H + OT_H - 1) / OT_H;
  int bid = blockIdx.x;
  const int b = bid / (tiles_w * tiles_h);
  bid -= b * tiles_w * tiles_h;
  const int th = bid / tiles_w, tw = bid - th * tiles_w;
  const int h0 = th * OT_H, w0 = tw * OT_W;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int NPIX = (OT_H + 2) * (OT_W + 2);
  for (int c0 = 0; c0 < C; c0 += OT_CK) {
    __syncthreads();
    for (int i = tid; i < NPIX * 4; i += 256) {
      const int pix = i >> 2, q = i & 3;
      const int py = pix / (OT_W + 2), px = pix - py * (OT_W + 2);
      const int ih = h0 + py - 1, iw = w0 + px - 1;
      const int c = c0 + q * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < C && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
        v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + ih) * W + iw) * C + c);
        const float* s = ss + ((size_t)b * C + c) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_s(fmaf(v[e], s[2 * e], s[2 * e + 1]));
      }
      *reinterpret_cast<f32x4*>(tile + pix * OT_LD + q * 4) = v;
    }
    __syncthreads();
    // The filter taps are the same for every lane: they are read through workgroup-uniform addresses, i.e. scalar loads into
    // SGPRs that feed the FMAs directly (round 3; the LDS copy of the weights cost 4 broadcast ds_read_b128 per tile read and
    // made the kernel LDS-instruction bound: 62 -> measured in profiles/archive/r03*).  Same FMA order as before: (tap, quad, element).
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* tp = tile + ((ty + tap / 3) * (OT_W + 2) + tx + tap % 3) * OT_LD;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (FULL || c0 + q * 4 < C) {                             // uniform
          const f32x4 v = *reinterpret_cast<const f32x4*>(tp + q * 4);
#pragma unroll
          for (int co = 0; co < COUT; ++co) {
            const float* wr = w + ((size_t)co * 9 + tap) * C + c0 + q * 4;
            acc[co] = fmaf(v[0], wr[0], acc[co]);
            acc[co] = fmaf(v[1], wr[1], acc[co]);
            acc[co] = fmaf(v[2], wr[2], acc[co]);
            acc[co] = fmaf(v[3], wr[3], acc[co]);
          }
        }
      }
    }
  }
  const int oh = h0 + ty, ow = w0 + tx;
  int t = 0;
  float ca = 0.f, cbb = 0.f, c1 = 0.f, c2 = 0.f, sg = 0.f;
  if (FUSE) {
    t = f.step_cur[0];
    ca = f.tb.a[t]; cbb = f.tb.b[t]; c1 = f.tb.c1[t]; c2 = f.tb.c2[t]; sg = f.tb.sigma[t];
    if (blockIdx.x == 0 && tid == 0) f.step_next[0] = t - 1;       // (nobody reads this slot before the next step's first kernel)
  }
  if (oh < H && ow < W) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const size_t idx = (((size_t)b * Cout + co) * H + oh) * W + ow;
      const float e = acc[co] + (bias ? bias[co] : 0.f);
      if (!FUSE || out) out[idx] = e;
      if (FUSE) {
        const float xv = f.x[idx], zv = f.z ? f.z[idx] : 0.f;
        float x0 = sub_rn(mul_rn(ca, xv), mul_rn(cbb, e));
        if (f.clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
        const float mean = add_rn(mul_rn(c1, x0), mul_rn(c2, xv));
        f.x[idx] = add_rn(mean, mul_rn(zv, sg));
      }
    }
  }
}

int conv_out_nchw(const float* x, const float* ss, int B, int H, int W, int C, const float* w, const float* bias,
                  int Cout, float* out_nchw, hipStream_t st, const StepFuse* fuse) {
  if (Cout > 4 || Cout < 1) { set_error("conv_out: Cout %d > 4 unsupported", Cout); return SR3_E_UNSUPPORTED; }
  if (C & 3) { set_error("conv_out: C %% 4 != 0"); return SR3_E_UNSUPPORTED; }
  if (!fuse && !out_nchw) { set_error("conv_out: null output"); return SR3_E_BADARG; }
  const int tiles = ((W + OT_W - 1) / OT_W) * ((H + OT_H - 1) / OT_H) * B;
  StepFuse f;
  memset(&f, 0, sizeof(f));
  if (fuse) f = *fuse;
#define SR3_CO_LAUNCH3(N, FU, FS)                                                                                            \
  hipLaunchKernelGGL((k_conv_out_nchw<N, FU, FS>), dim3(tiles), dim3(256), 0, st, x, ss, B, H, W, C, w, bias, Cout, out_nchw, f);
#define SR3_CO_LAUNCH(N)                                                                                                    \
  {                                                                                                                          \
    if (C % OT_CK == 0) { if (fuse) { SR3_CO_LAUNCH3(N, true, true) } else { SR3_CO_LAUNCH3(N, true, false) } }             \
    else { if (fuse) { SR3_CO_LAUNCH3(N, false, true) } else { SR3_CO_LAUNCH3(N, false, false) } }                          \
  }
  switch (Cout) {
    case 1: SR3_CO_LAUNCH(1) break;
    case 2: SR3_CO_LAUNCH(2) break;
    case 3: SR3_CO_LAUNCH(3) break;
    default: SR3_CO_LAUNCH(4) break;
  }
#undef SR3_CO_LAUNCH
#undef SR3_CO_LAUNCH3
  SR3_LAUNCH_CHECK("k_conv_out_nchw");
  return SR3_OK;
}

// ---------------------------------------------------------------------------------------------
// Embedding: PositionalEncoding / TimeEmbedding -> Linear -> Swish -> Linear (unet.py:18-31,
// 179-184; ddpm unet.py:19-34,165-170), then every per-block projection at once:
// FeatureWiseAffine Linear(inner -> Cout) (sr3 unet.py:34-50) or Swish -> Linear (ddpm :81-84).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_embed(const EmbedParams p) {
  extern __shared__ f32x4 smem_v[];
  float* enc = reinterpret_cast<float*>(smem_v);           // [inner]
  float* hid = enc + p.inner;                              // [4*inner]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int half = p.inner / 2;
  float lv;
  if (p.variant == 0) {
    lv = p.step_dev ? p.level_table[p.step_dev[0] + 1] : p.level[b];
  } else {
    lv = p.step_dev ? (float)p.step_dev[0] : (float)p.tstep[b];
  }
  if (p.step_out && b == 0 && tid == 0) p.step_out[0] = p.step_dev[0];      // (sr3_reverse_step: the slot the step's tail reads)
  for (int k = tid; k < half; k += 256) {
    const float arg = lv * p.freq[k];
    enc[k] = sinf(arg);
    enc[half + k] = cosf(arg);
  }
  __syncthreads();
  const int hdim = 4 * p.inner;
  for (int j = tid; j < hdim; j += 256) {
    float s = p.b1[j];
    const float* wr = p.w1 + (size_t)j * p.inner;
    for (int k = 0; k < p.inner; ++k) s = fmaf(wr[k], enc[k], s);
    hid[j] = silu_s(s);
  }
  __syncthreads();
  for (int j = tid; j < p.inner; j += 256) {
    float s = p.b2[j];
    const float* wr = p.w2 + (size_t)j * hdim;
    for (int k = 0; k < hdim; ++k) s = fmaf(wr[k], hid[k], s);
    p.temb[(size_t)b * p.inner + j] = (p.variant == 1) ? silu_s(s) : s;
  }
}

__global__ __launch_bounds__(256) void k_film(const EmbedParams p) {
  extern __shared__ f32x4 smem_v[];
  float* e = reinterpret_cast<float*>(smem_v);
  const int b = blockIdx.y;
  for (int k = threadIdx.x; k < p.inner; k += 256) e[k] = p.temb[(size_t)b * p.inner + k];
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p.F) return;
  float s = p.bf[j];
  const float* wr = p.wf + (size_t)j * p.inner;
  for (int k = 0; k < p.inner; ++k) s = fmaf(wr[k], e[k], s);
  p.film[(size_t)b * p.F + j] = s;
}

int embed_forward(const EmbedParams& p, hipStream_t st) {
  if (p.inner & 1) { set_error("embed: inner must be even"); return SR3_E_UNSUPPORTED; }
  hipLaunchKernelGGL(k_embed, dim3(p.B), dim3(256), (size_t)5 * p.inner * sizeof(float), st, p);
  SR3_LAUNCH_CHECK("k_embed");
  if (p.F > 0) {
    hipLaunchKernelGGL(k_film, dim3((p.F + 255) / 256, p.B), dim3(256), (size_t)p.inner * sizeof(float), st, p);
    SR3_LAUNCH_CHECK("k_film");
  }
  return SR3_OK;
}

// ---------------------------------------------------------------------------------------------
// Fused reverse step (sr3 diffusion.py:141-149,162-174; ddpm :151-198):
//   x0 = a_t x - b_t eps ; clamp[-1,1] ; mean = c1_t x0 + c2_t x ; x <- mean + sigma_t z
// sigma_t = exp(0.5 logvar_t) with sigma_0 := 0 (the reference's `t > 0` branch / nonzero_mask).
// Each product/sum is rounded separately (no contraction) so the update is bit-identical to the
// reference's elementwise torch ops for the same eps.  t comes from a device counter (graph
// replay), a per-sample int64 array (DDPM API) or the host.
// ---------------------------------------------------------------------------------------------
template <bool CLIP>
__global__ __launch_bounds__(256) void k_p_sample_update(float* __restrict__ x, const float* __restrict__ eps,
                                                          const float* __restrict__ z, StepTables tb,
                                                          const int* __restrict__ step_dev,
                                                          const int64_t* __restrict__ tps, int step_host,
                                                          int per_image, size_t total4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e0 = i * 4;
    const int b = (int)(e0 / per_image);
    const int t = step_dev ? step_dev[0] : (tps ? (int)tps[b] : step_host);
    const float a = tb.a[t], bb = tb.b[t], c1 = tb.c1[t], c2 = tb.c2[t], sg = tb.sigma[t];
    f32x4 xv = *reinterpret_cast<const f32x4*>(x + e0);
    const f32x4 ev = *reinterpret_cast<const f32x4*>(eps + e0);
    f32x4 zv = {0.f, 0.f, 0.f, 0.f};
    if (z) zv = *reinterpret_cast<const f32x4*>(z + e0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float x0 = sub_rn(mul_rn(a, xv[k]), mul_rn(bb, ev[k]));
      if (CLIP) x0 = fminf(fmaxf(x0, -1.f), 1.f);          // clip_denoised (sr3 diffusion.py:162-163)
      const float mean = add_rn(mul_rn(c1, x0), mul_rn(c2, xv[k]));
      xv[k] = add_rn(mean, mul_rn(zv[k], sg));
    }
    *reinterpret_cast<f32x4*>(x + e0) = xv;
  }
}

int p_sample_update(float* x, const float* eps, const float* z, StepTables tb, const int* step_dev,
                    const int64_t* t_per_sample, int step_host, int B, int per_image, hipStream_t st, bool clip) {
  if (per_image & 3) { set_error("p_sample_update: per-image size %% 4 != 0"); return SR3_E_UNSUPPORTED; }
  const size_t total4 = (size_t)B * per_image / 4;
  int blocks = (int)((total4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (clip) hipLaunchKernelGGL(k_p_sample_update<true>, dim3(blocks), dim3(256), 0, st, x, eps, z, tb, step_dev, t_per_sample,
                               step_host, per_image, total4);
  else hipLaunchKernelGGL(k_p_sample_update<false>, dim3(blocks), dim3(256), 0, st, x, eps, z, tb, step_dev, t_per_sample,
                          step_host, per_image, total4);
  SR3_LAUNCH_CHECK("k_p_sample_update");
  return SR3_OK;
}

__global__ void k_step_decrement(int* s) { if (threadIdx.x == 0 && blockIdx.x == 0) s[0] -= 1; }
int step_decrement(int* step_dev, hipStream_t st) {
  hipLaunchKernelGGL(k_step_decrement, dim3(1), dim3(64), 0, st, step_dev);
  SR3_LAUNCH_CHECK("k_step_decrement");
  return SR3_OK;
}

// q_sample: out = ca[b] * x0 + cb[b] * z   (sr3 diffusion.py:212-219, ddpm :259-267)
__global__ __launch_bounds__(256) void k_q_sample(const float* __restrict__ x0, const float* __restrict__ z,
                                                   const float* __restrict__ ca, const float* __restrict__ cb,
                                                   int per_image, size_t total4, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e0 = i * 4;
    const int b = (int)(e0 / per_image);
    const float a = ca[b], s = cb[b];
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x0 + e0);
    const f32x4 zv = *reinterpret_cast<const f32x4*>(z + e0);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = add_rn(mul_rn(a, xv[k]), mul_rn(s, zv[k]));
    *reinterpret_cast<f32x4*>(out + e0) = o;
  }
}

int q_sample(const float* x0, const float* z, const float* ca, const float* cb, int B, int per_image, float* out,
             hipStream_t st) {
  if (per_image & 3) { set_error("q_sample: per-image size %% 4 != 0"); return SR3_E_UNSUPPORTED; }
  const size_t total4 = (size_t)B * per_image / 4;
  int blocks = (int)((total4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_q_sample, dim3(blocks), dim3(256), 0, st, x0, z, ca, cb, per_image, total4, out);
  SR3_LAUNCH_CHECK("k_q_sample");
  return SR3_OK;
}

}  // namespace sr3
