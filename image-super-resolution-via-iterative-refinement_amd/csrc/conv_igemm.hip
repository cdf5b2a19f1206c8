// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (157 TF peak; bitwise an fmaf chain, see MI355X_MICROARCH.md).
//
// Replaces, on the reference hot path, every `nn.Conv2d` call of the UNet together with the
// GroupNorm -> Swish prologue of `Block` (model/sr3_modules/unet.py:80-91), the FiLM add
// (:34-50, :108), the residual add (:110, :142), the nearest upsample (:58-65), the stride-2
// Downsample (:68-74) and the skip `torch.cat` (:255) -- none of which is materialised.
//
// GEMM view:  out[m][n] = sum_k A[m][k] * W[n][k],  m = (b, oh, ow), n = cout, k = (tap, cin).
// Layout: activations NHWC (k contiguous for a fixed tap), weights OHWI ([n][tap][cin]), so both
// operands are "k-contiguous rows": a float4 global load lands as one ds_write_b128 into an
// LDS tile [rows][32 + 4 pad]; fragments come back as one ds_read_b128 per 4 MFMA k-steps
// (the lane's 4 consecutive k; lanes 0-31 / 32-63 take k-quads 0 / 1 of each 8-k group, which
// is a permutation of the k order shared by A and B, so the sum is unchanged).
// Block = 256 threads = 2x2 waves, tile BM x BN, k-step 32 channels of one filter tap,
// double-buffered LDS with register staging (global -> VGPR -> [GN/SiLU] -> LDS; the global loads
// run two k-steps ahead in two register sets), one barrier per k-step.  Split-K writes fp32 slabs
// that a second kernel reduces (deterministic).
//
// SPLIT instantiations (ConvParams::igemm_split; plan option `gemm_split`): the same loader, prologue and epilogue on the
// bf16 matrix pipe without losing bits -- every fp32 operand is written to LDS as three bf16 planes (x = h + m + l, 8 + 8 + 8
// significant bits, each residual exact in fp32) while it is staged, and a product is the six v_mfma_f32_32x32x16_bf16
// products hh + hm + mh + mm + hl + lh with fp32 accumulation (dropped terms <= 2^-24 of a product): 6/16 of the fp32
// instruction's matrix-pipe time (the split's VALU work is paid beside it in full: VALU instructions do not execute next to an
// MFMA on gfx950, profiles/r04g_mfma_overlap_bf16.txt).  LDS rows are 32 bf16
// (64 bytes, no padding), their four 16-byte segments XOR-swizzled with (row >> 2) & 3 (conv3x3_halo.hip's MODE 1 layout:
// 8-byte staging writes and 16-byte fragment reads both conflict-free).  The 128 x 128 tile single-buffers its 48 KB stage
// (two workgroups per CU; the second barrier of a k-step is covered by the other workgroup), the smaller tiles double-buffer.
#include <stdlib.h>

#include <type_traits>

#include "sr3_common.h"

namespace sr3 {

__device__ __forceinline__ float silu_f(float v) {
  // x * sigmoid(x); exp is the accurate libm one, reciprocal is v_rcp_f32 (1 ulp)
  return SR3_SILU(v);
}

// SPLIT: 0 = fp32 MFMA, 1 = 3 x bf16 split with both operands split while staged, 2 = the same with the WEIGHTS pre-split AND laid out
// as the MFMA's B fragments (ConvParams::w_split, written once per weight change by igemm_split_weights -- a plan keeps them in its
// derived buffer): a wave reads its B fragments straight from global memory, one coalesced 1 KB load per (n block, K = 16 step,
// plane), one k-step ahead in registers -- the weights never pass through LDS, and a workgroup stages and splits only its
// activation rows: half of the staging VALU work, half of the LDS stage (round 6; round 5's [plane][quad] layout still went
// through the LDS staging with three loads per quad and lost, profiles/r05c_*)
template <int BM, int BN, int TAPS, int SPLITM>
__global__ __launch_bounds__(256, 2) void k_conv_igemm(const ConvParams p) {
  constexpr bool SPLIT = SPLITM != 0, WPRE = SPLITM == 2;
  constexpr int BK = 32, LDK = 36, LDB = 32;
  constexpr int NST = (SPLIT && !WPRE && BM + BN > 192) ? 1 : 2;      // LDS stages (single-stage on the 64x64 split tile too: 7.988 vs 7.985 ms per step, no gain)
  constexpr int AR = BM / 32, BR = BN / 32;  // loader rows per thread
  constexpr int WM = BM / 2, WN = BN / 2;    // wave tile, 2x2 waves
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int STAGE = SPLIT ? (BM + (WPRE ? 0 : BN)) * LDB * 3 / 2 : (BM + BN) * LDK;     // floats per stage (WPRE: the A rows only)
  auto swz = [](int row, int seg) { return row * LDB + (((seg ^ (row >> 2)) & 3) << 3); };   // bf16 index of a 16-byte segment
  extern __shared__ f32x4 smem_v[];
  float* smem = reinterpret_cast<float*>(smem_v);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = tid & 7, lrow = tid >> 3;
  const int Cin = p.C0 + p.C1;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.B * HoWo;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n;
  const int tile_n = blockIdx.x - tile_m * tiles_n;
  const int Hi = p.Hs << p.ups, Wi = p.Ws << p.ups;
  constexpr int PAD = (TAPS == 9) ? 1 : 0;
  const int nchunks = (Cin + BK - 1) / BK;
  const int total = nchunks * TAPS;
  const int per = (total + p.ksplit - 1) / p.ksplit;
  const int it0 = blockIdx.y * per;
  const int it1 = min(total, it0 + per);

  // ---- per-thread loader rows -------------------------------------------------------------
  int rb[AR], rih[AR], riw[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = tile_m * BM + lrow + 32 * i;
    if (m < M) {
      const int b = m / HoWo;
      const int rem = m - b * HoWo;
      const int oh = rem / p.Wo;
      const int ow = rem - oh * p.Wo;
      rb[i] = b;
      rih[i] = oh * p.stride - PAD;
      riw[i] = ow * p.stride - PAD;
    } else {
      rb[i] = -1; rih[i] = 0; riw[i] = 0;
    }
  }

  // Two register sets: the global loads run TWO k-steps ahead of the MFMAs (set = k-step parity), so a load has a whole
  // compute + stage phase to land even when the operands are cold (weights straight from HBM inside a forward; the split
  // instantiations' k-steps are ~1/3 as long as the fp32 ones: with one step of prefetch they were 14-40 % slower inside the
  // forward than in isolation, profiles/r04f_gemm_split_sweep.txt)
  f32x4 ra[2][AR], rw[2][WPRE ? 1 : BR], ssa[AR], ssb[AR];       // (the GroupNorm pairs, L2-hot, stay one step ahead: one set)
  bf16x8 bfr[2][2][WPRE ? NI : 1][3];                            // WPRE: B fragments [set][K = 16 step][n block][plane], one k-step ahead
  bool aok[2][AR], wok[2][BR];
  int aoff[2][AR];          // element offset of the staged quad (dropout mask index)
  using SET0 = std::integral_constant<int, 0>;
  using SET1 = std::integral_constant<int, 1>;

  // Every global load below is UNCONDITIONAL (out-of-range lanes read element 0 of the same
  // buffer) and the zero mask is applied when the registers are written to LDS, after the MFMA
  // block: a predicated load would put each load in its own exec-masked branch and drain vmcnt
  // at the join, exposing the full memory latency every k-step.  32-bit element offsets (host
  // checks every tensor has < 2^31 elements).
  auto load_global = [&](auto set_tag, int it) {
    constexpr int S = decltype(set_tag)::value;
    const int chunk = it / TAPS;
    const int tap = it - chunk * TAPS;
    const int fr = (TAPS == 9) ? tap / 3 : 0;
    const int fs = (TAPS == 9) ? tap - fr * 3 : 0;
    const int c = chunk * BK + kq * 4;
    const bool cvalid = c < Cin;
    const int ce = cvalid ? c : 0;
    const bool second = ce >= p.C0;
    const float* sp = second ? p.src1 : p.src0;
    const int sC = second ? p.C1 : p.C0;
    const int cs = second ? ce - p.C0 : ce;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int ih = rih[i] + fr, iw = riw[i] + fs;
      const bool ok = cvalid && rb[i] >= 0 && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi;
      aok[S][i] = ok;
      const int pix = (rb[i] * p.Hs + (ih >> p.ups)) * p.Ws + (iw >> p.ups);
      const int off = ok ? pix * sC + cs : 0;
      aoff[S][i] = off;
      ra[S][i] = *reinterpret_cast<const f32x4*>(sp + off);
    }
    if constexpr (!WPRE) {
#pragma unroll
      for (int j = 0; j < BR; ++j) {
        const int n = tile_n * BN + lrow + 32 * j;
        const bool ok = cvalid && n < p.Cout;
        wok[S][j] = ok;
        const int off = ok ? (n * TAPS + tap) * Cin + c : 0;
        rw[S][j] = *reinterpret_cast<const f32x4*>(p.w + off);
      }
    }
  };
  // WPRE: the B fragments of k-step `it` for this wave's n blocks: w_split[n block][it][K = 16 step 2][plane 3][lane 64][8 bf16]
  // (igemm_split_weights; zero outside Cout / Cin, so no bounds here)
  const int wave_n_ = (tid >> 6) & 1;
  auto load_bfrag = [&](auto set_tag, int it) {
    constexpr int S = decltype(set_tag)::value;
    if constexpr (WPRE) {
      const int nb0 = (tile_n * BN + wave_n_ * WN) >> 5;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const bf16x8* q = reinterpret_cast<const bf16x8*>(p.w_split) + ((size_t)(nb0 + j) * total + it) * (2 * 3 * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) bfr[S][ks][j][pl] = q[(ks * 3 + pl) * 64];
      }
    }
  };
  auto load_ss = [&](int it) {       // scale / shift pairs of the k-step that is staged next
    const int c = (it / TAPS) * BK + kq * 4;
    const int ce = c < Cin ? c : 0;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int be = rb[i] >= 0 ? rb[i] : 0;
      const float* q = p.ss + (be * Cin + ce) * 2;
      ssa[i] = *reinterpret_cast<const f32x4*>(q);
      ssb[i] = *reinterpret_cast<const f32x4*>(q + 4);
    }
  };

  auto store_lds = [&](auto set_tag, int stage) {
    constexpr int S = decltype(set_tag)::value;
    float* A = smem + stage * STAGE;
    float* Bw = A + BM * LDK;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      f32x4 v = ra[S][i];
      if (p.act != 0) {
        v.x = fmaf(v.x, ssa[i].x, ssa[i].y);
        v.y = fmaf(v.y, ssa[i].z, ssa[i].w);
        v.z = fmaf(v.z, ssb[i].x, ssb[i].y);
        v.w = fmaf(v.w, ssb[i].z, ssb[i].w);
        if (p.act == 2) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
        if (p.drop_thresh != 0) {
          const unsigned i0 = (unsigned)aoff[S][i];
          v.x *= drop_mask(p.drop_seed, i0, p.drop_thresh, p.drop_scale);
          v.y *= drop_mask(p.drop_seed, i0 + 1, p.drop_thresh, p.drop_scale);
          v.z *= drop_mask(p.drop_seed, i0 + 2, p.drop_thresh, p.drop_scale);
          v.w *= drop_mask(p.drop_seed, i0 + 3, p.drop_thresh, p.drop_scale);
        }
      }
      v = aok[S][i] ? v : zero;     // zero padding is applied AFTER the activation, as the reference does
      if constexpr (SPLIT) {
        __bf16* Ab = reinterpret_cast<__bf16*>(A);            // planes [3][BM][LDB]
        bf16x4 h, m, l;
        split3(v, h, m, l);
        const int o = swz(lrow + 32 * i, kq >> 1) + (kq & 1) * 4;
        *reinterpret_cast<bf16x4*>(&Ab[o]) = h;
        *reinterpret_cast<bf16x4*>(&Ab[BM * LDB + o]) = m;
        *reinterpret_cast<bf16x4*>(&Ab[2 * BM * LDB + o]) = l;
      } else {
        *reinterpret_cast<f32x4*>(&A[(lrow + 32 * i) * LDK + kq * 4]) = v;
      }
    }
    if constexpr (!WPRE)
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      const f32x4 v = wok[S][j] ? rw[S][j] : zero;
      if constexpr (SPLIT) {
        __bf16* Bb = reinterpret_cast<__bf16*>(A) + 3 * BM * LDB;     // planes [3][BN][LDB]
        bf16x4 h, m, l;
        split3(v, h, m, l);
        const int o = swz(lrow + 32 * j, kq >> 1) + (kq & 1) * 4;
        *reinterpret_cast<bf16x4*>(&Bb[o]) = h;
        *reinterpret_cast<bf16x4*>(&Bb[BN * LDB + o]) = m;
        *reinterpret_cast<bf16x4*>(&Bb[2 * BN * LDB + o]) = l;
      } else {
        *reinterpret_cast<f32x4*>(&Bw[(lrow + 32 * j) * LDK + kq * 4]) = v;
      }
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int arow = wave_m * WM + (lane & 31);
  const int brow = wave_n * WN + (lane & 31);
  const int kh = (lane >> 5) * 4;

  auto compute = [&](int stage, auto bset_tag) {
    [[maybe_unused]] constexpr int BS = decltype(bset_tag)::value;
    const float* A = smem + stage * STAGE;
    const float* Bw = A + BM * LDK;
    if constexpr (SPLIT) {
      const __bf16* Ab = reinterpret_cast<const __bf16*>(A);
      const __bf16* Bb = Ab + 3 * BM * LDB;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {              // two K = 16 steps per 32-channel k-step
        const int seg = ks * 2 + (lane >> 5);       // 16-byte segment (8 channels) of the row
        bf16x8 a[MI][3], b[NI][3];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int o = swz(arow + 32 * i, seg);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) a[i][pl] = *reinterpret_cast<const bf16x8*>(&Ab[pl * BM * LDB + o]);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if constexpr (WPRE) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[j][pl] = bfr[BS][ks][j][pl];
          } else {
            const int o = swz(brow + 32 * j, seg);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[j][pl] = *reinterpret_cast<const bf16x8*>(&Bb[pl * BN * LDB + o]);
          }
        }
        // product-major order (independent accumulators between dependent MFMAs), smallest terms first
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f32x4*>(&A[(arow + 32 * i) * LDK + kk * 8 + kh]);
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const f32x4*>(&Bw[(brow + 32 * j) * LDK + kk * 8 + kh]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
    }
  };

  // main loop, unrolled by two so the register set of a k-step is a compile-time constant: at the top of step i the LDS holds
  // step i, set (i + 1) & 1 holds step i + 1 (loaded one step ago) and set i & 1 is free for step i + 2
  const int nsteps = it1 - it0;
  auto step = [&](auto set_cur, auto set_next, int i, int cur) {
    if (i + 2 < nsteps && !(p.dbg & 6)) load_global(set_cur, it0 + i + 2);
    load_bfrag(set_next, min(it0 + i + 1, it1 - 1));       // (unconditional: the last step re-fetches its own fragments)
    if (i + 1 < nsteps && p.act != 0) load_ss(it0 + i + 1);
    if (!(p.dbg & 1)) compute(cur, set_cur);
    if constexpr (NST == 1) __syncthreads();          // single stage: every wave has read its fragments
    if (i + 1 < nsteps && !(p.dbg & 10)) store_lds(set_next, NST == 2 ? cur ^ 1 : 0);
    __syncthreads();
  };
  if (nsteps > 0) {
    load_global(SET0{}, it0);
    load_bfrag(SET0{}, it0);
    if (nsteps > 1) load_global(SET1{}, it0 + 1);
    if (p.act != 0) load_ss(it0);
    store_lds(SET0{}, 0);
    __syncthreads();
    for (int i = 0; i < nsteps; i += 2) {
      step(SET0{}, SET1{}, i, 0);
      if (i + 1 < nsteps) step(SET1{}, SET0{}, i + 1, NST == 2 ? 1 : 0);
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------
  // D layout of the 32x32 MFMA: reg r of lane l -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31.
  // Rows are walked in groups of 8 (g = r>>2): when Ho*Wo % 8 == 0 a group never straddles an
  // image, so the image index (FiLM row, statistics bucket) is wave-uniform per group.
  const bool direct = (p.ksplit == 1);
  const bool grp_uniform = (HoWo & 7) == 0;
  float* dst = direct ? p.out : p.partial + (size_t)blockIdx.y * M * p.Cout;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = tile_n * BN + wave_n * WN + 32 * j + (lane & 31);
    const bool nok = n < p.Cout;
    float bn = 0.f;
    if (direct && nok && p.bias) bn = p.bias[n];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int mbase = tile_m * BM + wave_m * WM + 32 * i + 8 * g;   // wave-uniform
        const int bg = mbase / HoWo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = mbase + e + 4 * (lane >> 5);
          if (m < M && nok) {
            float v = acc[i][j][g * 4 + e];
            if (direct) {
              const int b = grp_uniform ? bg : m / HoWo;
              v += bn;
              if (p.film) v += p.film[(size_t)b * p.film_stride + n];
              if (p.res0)
                v += (n < p.RC0) ? p.res0[(size_t)m * p.RC0 + n] : p.res1[(size_t)m * p.RC1 + (n - p.RC0)];
            }
            dst[(size_t)m * p.Cout + n] = v;
          }
        }
      }
    }
  }
}

// ---- split-K reduction + epilogue (+ optional partial GroupNorm statistics of the output) -------
// A block owns `rpb` consecutive output rows (pixels) -- inside one image when statistics are
// requested -- lanes run along channel quads (coalesced 16-byte accesses), the 4 waves along rows.
__global__ __launch_bounds__(256) void k_splitk_reduce(const ConvParams p, int rpb) {
  __shared__ double red[4 * 64 * 8];
  const int HoWo = p.Ho * p.Wo;
  const size_t M = (size_t)p.B * HoWo;
  const int nq = p.Cout >> 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row0 = (size_t)blockIdx.x * rpb;
  for (int q0 = 0; q0 < nq; q0 += 64) {
    const int q = q0 + lane;
    const int n = q * 4;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (q < nq) {
      f32x4 cb = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) cb += *reinterpret_cast<const f32x4*>(p.bias + n);
      if (p.x2_w && p.x2_bias) cb += *reinterpret_cast<const f32x4*>(p.x2_bias + n);
      for (int r = wave; r < rpb; r += 4) {
        const size_t m = row0 + r;
        if (m >= M) break;
        f32x4 v = cb;
        // four slabs in flight (round 6; one load per iteration made the wave wait for each slab in turn); same order of additions
        auto slab = [&](int s) { return *reinterpret_cast<const f32x4*>(p.partial + ((size_t)s * M + m) * p.Cout + n); };
        int s = 0;
        for (; s + 4 <= p.ksplit; s += 4) {
          const f32x4 a0 = slab(s), a1 = slab(s + 1), a2 = slab(s + 2), a3 = slab(s + 3);
          v += a0; v += a1; v += a2; v += a3;
        }
        for (; s < p.ksplit; ++s) v += slab(s);
        const int b = (int)(m / HoWo);
        if (p.film) v += *reinterpret_cast<const f32x4*>(p.film + (size_t)b * p.film_stride + n);
        if (p.res0) {
          if (n < p.RC0) v += *reinterpret_cast<const f32x4*>(p.res0 + m * p.RC0 + n);
          else v += *reinterpret_cast<const f32x4*>(p.res1 + m * p.RC1 + (n - p.RC0));
        }
        *reinterpret_cast<f32x4*>(p.out + m * p.Cout + n) = v;
        if (p.ostat) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const double dv = (double)v[e]; s1[e] += dv; s2[e] += dv * dv; }
        }
      }
    }
    if (p.ostat) {      // uniform branch: fixed-order reduction over the 4 row lanes, one plain store
#pragma unroll
      for (int e = 0; e < 4; ++e) { red[(wave * 64 + lane) * 8 + e] = s1[e]; red[(wave * 64 + lane) * 8 + 4 + e] = s2[e]; }
      __syncthreads();
      if (wave == 0 && q < nq && row0 < M) {
        const int b = (int)(row0 / HoWo);
        const int T = HoWo / rpb;
        const int slice = (int)(row0 - (size_t)b * HoWo) / rpb;
        double* o = p.ostat + (((size_t)b * T + slice) * p.Cout + n) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          double a1 = 0.0, a2 = 0.0;
#pragma unroll
          for (int w = 0; w < 4; ++w) { a1 += red[(w * 64 + lane) * 8 + e]; a2 += red[(w * 64 + lane) * 8 + 4 + e]; }
          o[2 * e] = a1; o[2 * e + 1] = a2;
        }
      }
      __syncthreads();
    }
  }
}

// ---- pre-split weights of the SPLIT instantiations (ConvParams::w_split) --------------------------------------------------------
// out[n block][it = chunk * taps + tap][K = 16 step 2][plane 3][lane 64][8 bf16]: lane l of a fragment holds W[n = 32 nb + (l & 31)]
// [tap][c = 32 chunk + 16 ks + 8 (l >> 5) .. + 7] -- the B operand of v_mfma_f32_32x32x16_bf16 in the k order the kernel's A
// fragments have (plain channel order inside a 32-channel k-step), zero outside Cout / Cin.  One thread per fragment lane.
__global__ __launch_bounds__(256) void k_split_weights(const float* __restrict__ w, int Cout, int taps, int Cin, int nchunks, long nfrag,
                                                        __bf16* __restrict__ out) {
  for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < nfrag; f += (long)gridDim.x * blockDim.x) {
    const int l = (int)(f & 63);
    long r = f >> 6;
    const int ks = (int)(r & 1); r >>= 1;
    const int it = (int)(r % ((long)nchunks * taps));
    const int nb = (int)(r / ((long)nchunks * taps));
    const int chunk = it / taps, tap = it - chunk * taps;
    const int n = nb * 32 + (l & 31), c = chunk * 32 + ks * 16 + 8 * (l >> 5);
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
    if (n < Cout) {
      const float* q = w + ((size_t)n * taps + tap) * Cin + c;
      if (c < Cin) lo = *reinterpret_cast<const f32x4*>(q);
      if (c + 4 < Cin) hi = *reinterpret_cast<const f32x4*>(q + 4);
    }
    bf16x8 h, m, lw;
    split3x8(lo, hi, h, m, lw);
    bf16x8* o = reinterpret_cast<bf16x8*>(out) + (((size_t)nb * nchunks * taps + it) * 2 + ks) * (3 * 64) + l;
    o[0] = h; o[64] = m; o[128] = lw;
  }
}
size_t igemm_wsplit_floats(int Cout, int taps, int Cin) {       // 3 planes x 2 bytes per padded weight
  return (size_t)((Cout + 31) / 32) * 32 * taps * ((Cin + 31) / 32) * 32 * 6 / 4;
}
int igemm_split_weights(const float* w, int Cout, int taps, int Cin, float* out, hipStream_t st) {
  if (Cin & 3) { set_error("conv: Cin %% 4 != 0"); return SR3_E_UNSUPPORTED; }
  const int nchunks = (Cin + 31) / 32;
  const long nfrag = (long)((Cout + 31) / 32) * nchunks * taps * 2 * 64;
  int blocks = (int)((nfrag + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_split_weights, dim3(blocks), dim3(256), 0, st, w, Cout, taps, Cin, nchunks, nfrag, reinterpret_cast<__bf16*>(out));
  SR3_LAUNCH_CHECK("k_split_weights");
  return SR3_OK;
}

// ---- host side --------------------------------------------------------------------------------
namespace {
struct TileCfg { int bm, bn; };
const TileCfg kCfgs[5] = {{0, 0}, {128, 128}, {128, 64}, {64, 64}, {64, 128}};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

template <int BM, int BN, int TAPS, int SPLIT = 0>
int launch_conv(const ConvParams& p, hipStream_t st) {
  static std::atomic<uint64_t> attr_done{0};
  constexpr int smem = SPLIT == 2 ? 2 * BM * 192 : SPLIT ? ((BM + BN > 192) ? 1 : 2) * (BM + BN) * 192 : 2 * (BM + BN) * 36 * 4;
  auto kern = k_conv_igemm<BM, BN, TAPS, SPLIT>;
  if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), smem, attr_done)) return rc;
  const int M = p.B * p.Ho * p.Wo;
  dim3 grid(cdiv(M, BM) * cdiv(p.Cout, BN), p.ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
  SR3_LAUNCH_CHECK("k_conv_igemm");
  return SR3_OK;
}
}  // namespace

static thread_local hipEvent_t g_mid_event = nullptr;
void conv_set_mid_event(hipEvent_t ev) { g_mid_event = ev; }

void conv_pick(const ConvParams& p, int& tile_cfg, int& ksplit) {
  const int M = p.B * p.Ho * p.Wo;
  const int Cin = p.C0 + p.C1;
  const int taps = p.ksize * p.ksize;
  const int nchunks = cdiv(Cin, 32);
  static const bool no_wide = getenv("SR3_NO_WIDE") != nullptr;     // A/B knob: never pick the 8-wave 256x128 tile
  if (tile_cfg == 0) {
    HaloGeom g;
    if (p.ksize == 3 && p.stride == 1) {
      if (p.Cout <= 64 && halo_geometry(p, 6, &g)) tile_cfg = 6;
      else if (p.Cout > 64 && !no_wide && halo_geometry(p, 9, &g) &&
               (long)cdiv(p.Cout, 128) * g.tiles_w * g.tiles_h * cdiv(p.B, g.NB) >= 256)
        tile_cfg = 9;     // 8-wave 256x128 tile: half the weight traffic, when it still fills every CU
      else if (halo_geometry(p, 5, &g)) tile_cfg = 5;
    }
  }
  // 3 x bf16 split instantiations: the 64x64 tile as well.  In isolation (warm operands) the larger split tiles win on most
  // layer shapes (128x128: qkv 71 vs 94 us), but INSIDE the forward -- activations just written by the previous kernel, weights
  // from HBM -- they lose by up to 2.5x on the short-K layers (one round of 384-512 workgroups loads, computes and stores in
  // lock step), and per-launch timing of the whole forward with every tile forced in turn (plan option gemm_tile,
  // profiles/r04f_gemm_split_sweep.txt) gives 64x64 2.31 / 1.96 / 1.59 / 2.02 ms for tiles 1-4 (fp32 MFMA: 2.59 / 2.27 / 1.86 / 2.34)
  if (tile_cfg == 0 && p.igemm_split && (p.ksize == 1 || p.stride == 2)) tile_cfg = 3;
  if (tile_cfg == 0 && p.ksize == 1) tile_cfg = 3;   // 1x1 convs: the 64x64 tile measured fastest on every layer shape of the
                                                     // BASELINE networks (76-81 vs 58-64 TF at 16x16, 78 vs 69 at 128x128)
  if (tile_cfg == 0) {
    // im2col kernel: largest tile that still gives >= ~2 workgroups per CU
    const int order_wide[3] = {1, 2, 3};
    const int order_narrow[2] = {2, 3};
    const int* order = p.Cout > 64 ? order_wide : order_narrow;
    const int norder = p.Cout > 64 ? 3 : 2;
    tile_cfg = order[norder - 1];
    for (int k = 0; k < norder; ++k) {
      const TileCfg c = kCfgs[order[k]];
      if ((long)cdiv(M, c.bm) * cdiv(p.Cout, c.bn) >= 448) { tile_cfg = order[k]; break; }
    }
  }
  if (tile_cfg == 22) {
    // 1x1 GEMM kernel (gemm1x1.hip; two workgroups per CU): split K below one round of workgroups, >= 4 k-steps (128 channels) per split
    if (ksplit == 0) {
      const long tiles = (long)(M / gemm1x1_rows(p)) * (p.Cout / gemm1x1_cols(p));
      int ks = 1;
#ifndef SR3_G1_SPLIT_BELOW
#define SR3_G1_SPLIT_BELOW 128
#endif
#ifndef SR3_G1_S2_SPLIT_BELOW
#define SR3_G1_S2_SPLIT_BELOW 512
#endif
      // the stride-2 form walks 9 k-steps per 32-channel chunk: K is long (>= 36 steps at 128 channels) and the maps are small, so it
      // splits whenever the tiles do not fill the 512 slots, >= 8 k-steps per split
      const int ksteps = nchunks * taps;
      const int below = taps == 9 ? SR3_G1_S2_SPLIT_BELOW : SR3_G1_SPLIT_BELOW;
      if (tiles < below && tiles > 0) {
        ks = (int)((512 + tiles - 1) / tiles);
        const int cap = taps == 9 ? (ksteps / 8 > 1 ? ksteps / 8 : 1) : (nchunks / 4 > 1 ? nchunks / 4 : 1);
        if (ks > cap) ks = cap;
        if (ks > 16) ks = 16;
      }
      while (ks > 1 && (long)(ks - 1) * cdiv(ksteps, ks) >= ksteps) --ks;
      ksplit = ks;
    }
    return;
  }
  if (ksplit == 0 && tile_cfg == 11) {
    // Winograd kernel: one 8-wave workgroup per CU, so one full round of 256 is the target (512 for the four-wave workgroups of
    // conv3x3_wino2.hip, two per CU); a split keeps >= 4 chunks (64 input channels)
    WinoGeom wg;
    if (!wino_geometry(p, &wg)) { ksplit = 1; return; }
    const long tiles = wino_workgroups(p, wg);
    const int units = wino_chunks(p);
    const int maxck = wino_max_chunks_per_split(wg);
    const long round = p.wino_split == 2 ? 512 : 256;
    // (tried in round 6: no split from half a round on for the 8 x 16 tile -- 11 reduce launches fewer, the convs 0.13-0.24 ms slower:
    // a wash, profiles/r06_wino2_experiments.txt)
    int ks = 1;
    if (tiles < round) {
      ks = (int)((round + tiles - 1) / tiles);
      const int cap = units / 4 > 1 ? units / 4 : 1;
      if (ks > cap) ks = cap;
      if (ks > 16) ks = 16;
    }
    if (wg.NB != 1 && ks < 2 && units >= 2) ks = 2;          // the four-image tile (8 x 8 maps) has no direct epilogue
    if (cdiv(units, ks) > maxck) ks = cdiv(units, maxck);    // chunks per split the kernel's LDS block of GroupNorm pairs holds
    while (ks > 1 && (long)(ks - 1) * cdiv(units, ks) >= units) --ks;
    ksplit = ks;
    return;
  }
  if (ksplit == 0) {
    long tiles;
    int units, min_units;
    if (tile_cfg >= 5) {
      HaloGeom g;
      if (!halo_geometry(p, tile_cfg, &g)) { ksplit = 1; return; }
      const int bn = halo_cfg_bn(tile_cfg);
      tiles = (long)cdiv(p.Cout, bn) * g.tiles_w * g.tiles_h * cdiv(p.B, g.NB);
      units = nchunks; min_units = 2;
    } else {
      const TileCfg c = kCfgs[tile_cfg];
      tiles = (long)cdiv(M, c.bm) * cdiv(p.Cout, c.bn);
      units = nchunks * taps; min_units = 6;
    }
    int ks = 1;
    // split K until the grid gives ~2 workgroups per CU (in-run A/B on MI355X: threshold 384 -> 14.87 ms per
    // step, 256 -> 14.96 ms; SR3_KSPLIT_TILES overrides)
    static const long split_below_env = getenv("SR3_KSPLIT_TILES") ? atol(getenv("SR3_KSPLIT_TILES")) : 384;
    // the 8-wave tiles run one workgroup per CU: one full round of 256 is the target there
    const bool wide = halo_cfg_one_wg_per_cu(tile_cfg);
    const long split_below = wide ? 256 : split_below_env;
    if (tiles < split_below) {
      ks = (int)(((wide ? 256 : 512) + tiles - 1) / tiles);
      const int cap = units / min_units > 1 ? units / min_units : 1;
      if (ks > cap) ks = cap;
      if (ks > 16) ks = 16;
    }
    while (ks > 1 && (long)(ks - 1) * cdiv(units, ks) >= units) --ks;   // no empty split
    ksplit = ks;
  }
}

// rows per block of the split-K reduce kernel; with statistics it must divide Ho*Wo (0: cannot fuse)
int splitk_rows_per_block(const ConvParams& p, bool stats) {
  const long M = (long)p.B * p.Ho * p.Wo;
  const int HoWo = p.Ho * p.Wo;
  int rpb = 64;
  while (rpb > 4 && M / rpb < 1024) rpb >>= 1;
  if (stats) {
    while (rpb > 1 && HoWo % rpb) rpb >>= 1;
    if (HoWo % rpb || HoWo / rpb > 256) return 0;
  }
  return rpb;
}

size_t conv_splitk_bytes(const ConvParams& p, int tile_cfg, int ksplit) {
  conv_pick(p, tile_cfg, ksplit);
  if (ksplit <= 1) return 0;
  return (size_t)ksplit * p.B * p.Ho * p.Wo * p.Cout * sizeof(float);
}

int conv_forward(const ConvParams& pin, int tile_cfg, int ksplit, float* scratch, size_t scratch_bytes,
                 hipStream_t st) {
  ConvParams p = pin;
  const int Cin = p.C0 + p.C1;
  if ((p.C0 & 3) || (p.C1 & 3) || (p.Cout & 3)) { set_error("conv: channel counts must be multiples of 4 (C0=%d C1=%d Cout=%d)", p.C0, p.C1, p.Cout); return SR3_E_UNSUPPORTED; }
  if (p.ksize != 1 && p.ksize != 3) { set_error("conv: ksize %d unsupported", p.ksize); return SR3_E_UNSUPPORTED; }
  if (p.stride != 1 && p.stride != 2) { set_error("conv: stride %d unsupported", p.stride); return SR3_E_UNSUPPORTED; }
  if (p.act != 0 && !p.ss) { set_error("conv: act needs ss"); return SR3_E_BADARG; }
  if (p.drop_thresh != 0 && (p.C1 != 0 || p.ups != 0 || p.act == 0)) { set_error("conv: dropout needs a single-source, non-upsampled, activated input"); return SR3_E_UNSUPPORTED; }
  if (p.C1 > 0 && !p.src1) { set_error("conv: C1 > 0 needs src1"); return SR3_E_BADARG; }
  if (p.res0 && p.RC0 + p.RC1 != p.Cout) { set_error("conv: residual channels %d+%d != Cout %d", p.RC0, p.RC1, p.Cout); return SR3_E_BADARG; }
  const int pad = p.ksize / 2;
  const int Hi = p.Hs << p.ups, Wi = p.Ws << p.ups;
  if ((Hi + 2 * pad - p.ksize) / p.stride + 1 != p.Ho || (Wi + 2 * pad - p.ksize) / p.stride + 1 != p.Wo) {
    set_error("conv: output dims %dx%d inconsistent with input %dx%d k%d s%d", p.Ho, p.Wo, Hi, Wi, p.ksize, p.stride);
    return SR3_E_BADARG;
  }
  {
    const double lim = 2147483647.0;
    const double e_in = (double)p.B * p.Hs * p.Ws * (double)(p.C0 > p.C1 ? p.C0 : p.C1);
    const double e_w = (double)p.Cout * p.ksize * p.ksize * (double)Cin;
    const double e_ss = (double)p.B * Cin * 2.0;
    if (e_in >= lim || e_w >= lim || e_ss >= lim) { set_error("conv: tensor exceeds 2^31 elements (32-bit offsets)"); return SR3_E_UNSUPPORTED; }
  }
  conv_pick(p, tile_cfg, ksplit);
  p.ksplit = ksplit;
  if (p.ostat && ksplit == 1 && (tile_cfg < 5 || tile_cfg >= 22)) { set_error("conv: fused output statistics need the halo kernel or split-K"); return SR3_E_UNSUPPORTED; }
  if (p.ostat && ksplit > 1 && splitk_rows_per_block(p, true) == 0) { set_error("conv: Ho*Wo does not allow fused split-K statistics"); return SR3_E_UNSUPPORTED; }
  { static const char* e = getenv("SR3_CONV_DBG"); p.dbg = e ? atoi(e) : 0; }
  if (ksplit > 1) {
    const size_t need = (size_t)ksplit * p.B * p.Ho * p.Wo * p.Cout * sizeof(float);
    if (!scratch || scratch_bytes < need) { set_error("conv: split-K scratch too small (%zu < %zu)", scratch_bytes, need); return SR3_E_NOMEM; }
    p.partial = scratch;
  }
  int rc;
  const bool k3 = p.ksize == 3;
  if (p.x2_w && (tile_cfg < 5 || tile_cfg == 11 || tile_cfg >= 22 || p.ups)) { set_error("conv: the fused 1x1 segment needs the halo kernel without upsampling"); return SR3_E_UNSUPPORTED; }
  if (p.x2_w && ((p.x2_C0 & 3) || (p.x2_C1 & 3) || !p.x2_src0 || (p.x2_C1 > 0 && !p.x2_src1))) { set_error("conv: bad x2 segment"); return SR3_E_BADARG; }
  if (tile_cfg == 22) {
    rc = gemm1x1_forward(p, 2, st);
  } else if (tile_cfg == 11) {
    rc = conv3x3_wino_forward(p, p.wino_u, st);
  } else if (tile_cfg >= 5) {
    HaloGeom g;
    if (tile_cfg > 10 || !halo_geometry(p, tile_cfg, &g)) { set_error("conv: halo tile_cfg %d does not fit this problem", tile_cfg); return SR3_E_UNSUPPORTED; }
    const int nchunks = cdiv(Cin, 32);
    if ((long)(ksplit - 1) * cdiv(nchunks, ksplit) >= nchunks && ksplit > 1) { set_error("conv: ksplit %d leaves an empty split over %d chunks", ksplit, nchunks); return SR3_E_BADARG; }
    rc = conv3x3_halo_forward(p, tile_cfg, g, st);
  } else
  if (p.igemm_split && p.w_split) switch (tile_cfg) {
    case 1: rc = k3 ? launch_conv<128, 128, 9, 2>(p, st) : launch_conv<128, 128, 1, 2>(p, st); break;
    case 2: rc = k3 ? launch_conv<128, 64, 9, 2>(p, st) : launch_conv<128, 64, 1, 2>(p, st); break;
    case 3: rc = k3 ? launch_conv<64, 64, 9, 2>(p, st) : launch_conv<64, 64, 1, 2>(p, st); break;
    case 4: rc = k3 ? launch_conv<64, 128, 9, 2>(p, st) : launch_conv<64, 128, 1, 2>(p, st); break;
    default: set_error("conv: bad tile_cfg %d", tile_cfg); return SR3_E_BADARG;
  } else
  if (p.igemm_split) switch (tile_cfg) {
    case 1: rc = k3 ? launch_conv<128, 128, 9, 1>(p, st) : launch_conv<128, 128, 1, 1>(p, st); break;
    case 2: rc = k3 ? launch_conv<128, 64, 9, 1>(p, st) : launch_conv<128, 64, 1, 1>(p, st); break;
    case 3: rc = k3 ? launch_conv<64, 64, 9, 1>(p, st) : launch_conv<64, 64, 1, 1>(p, st); break;
    case 4: rc = k3 ? launch_conv<64, 128, 9, 1>(p, st) : launch_conv<64, 128, 1, 1>(p, st); break;
    default: set_error("conv: bad tile_cfg %d", tile_cfg); return SR3_E_BADARG;
  } else
  switch (tile_cfg) {
    case 1: rc = k3 ? launch_conv<128, 128, 9>(p, st) : launch_conv<128, 128, 1>(p, st); break;
    case 2: rc = k3 ? launch_conv<128, 64, 9>(p, st) : launch_conv<128, 64, 1>(p, st); break;
    case 3: rc = k3 ? launch_conv<64, 64, 9>(p, st) : launch_conv<64, 64, 1>(p, st); break;
    case 4: rc = k3 ? launch_conv<64, 128, 9>(p, st) : launch_conv<64, 128, 1>(p, st); break;
    default: set_error("conv: bad tile_cfg %d", tile_cfg); return SR3_E_BADARG;
  }
  if (rc) return rc;
  if (ksplit > 1) {
    if (g_mid_event) { SR3_HIP(hipEventRecord(g_mid_event, st)); g_mid_event = nullptr; }
    if (p.fold) return splitk_reduce_fold(p, *p.fold, st);      // the reduce per (image, consumer group) + the next op's GroupNorm fold
    const int rpb = splitk_rows_per_block(p, p.ostat != nullptr);
    const long M = (long)p.B * p.Ho * p.Wo;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((M + rpb - 1) / rpb)), dim3(256), 0, st, p, rpb);
    SR3_LAUNCH_CHECK("k_splitk_reduce");
  }
  return SR3_OK;
}

}  // namespace sr3
