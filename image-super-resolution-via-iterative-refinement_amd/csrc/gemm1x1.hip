// 1x1 stride-1 convolution as a plain GEMM on the 3 x bf16 split arithmetic (gfx950; round 6).
//
// Replaces, for the layers it fits, the im2col kernel's split instantiation (conv_igemm.hip) on the reference's
// `res_conv` (model/sr3_modules/unet.py:102-103,110), the attention block's `qkv` and `out` projections (:119-121,138-141)
// and their DDPM twins: out[m][n] = sum_k A[m][k] W[n][k] with m = (image, pixel) -- NHWC rows are already the GEMM's A rows,
// no im2col -- optional GroupNorm affine on A (ConvParams::act == 1), epilogue + bias + FiLM + residual as the im2col kernel.
//
// What is different from conv_igemm.hip's SPLITM = 1 / 2 forms (the measurements that led here: profiles/r04f_gemm_split_sweep.txt,
// r06_gemm_wpre_fragment_major.txt -- those kernels are bound by VALU work per MFMA: ~150-300 VALU instructions per 12 MFMAs):
//  * tile 32*MI (M) x 128 (N), MI = 2 (64 rows; 32 where that would leave workgroup slots empty: gemm1x1_rows), four waves side by side
//    along N: every wave owns ONE 32-column block and all the rows, so its B fragments (the pre-split weights in MFMA fragment order,
//    igemm_split_weights) are its own -- read straight from global memory into registers two k-steps ahead, no LDS, no redundant fetch
//    inside the workgroup;
//  * the A rows are loaded two k-steps ahead, split once per workgroup and written to LDS in FRAGMENT order ([K = 16 step][m block]
//    [plane][lane][8 bf16]): a fragment read is base + lane * 16, conflict-free without a swizzle; one barrier per k-step, two stages;
//  * per k-step and wave: MI * 12 MFMAs against 4 * MI split elements per lane: ~3 plain VALU per MFMA, under the ~5 an MFMA hides when
//    they are interleaved (profiles/r05a_mfma_fillers.txt) -- the split is the `v_dot2c`-free one here (plain VALU only) and the staging
//    arithmetic is cut into slices placed by hand between the MFMAs (sched_barrier(0) pins every (MFMA, slice) group; see `step`);
//  * loop state is pointer increments (no division, no bounds: the host checks M % BM == 0, Cout % 128 == 0, channels % 32 == 0).
// What bounds it, and the scheduling variants that changed nothing: profiles/r06_gemm1x1.txt, DESIGN.md section 3.1g.
#include <stdlib.h>

#include "sr3_common.h"

#pragma clang diagnostic ignored "-Wpass-failed"      // (the slice loops of `step` are unrolled by a later pass than the one that warns)

namespace sr3 {

namespace {

// plain-VALU split (no v_dot2c: that one does not hide beside an MFMA): x = h + m + l, each residual exact
__device__ __forceinline__ float opaque(float v) { asm("" : "+v"(v)); return v; }      // (keeps the SLP vectoriser from pairing the
                                                                                        // subtractions into v_pk_add_f32, which does not hide)
__device__ __forceinline__ void split3_plain(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  bf16x2 hh;
  hh[0] = (__bf16)x0; hh[1] = (__bf16)x1;
  h = __builtin_bit_cast(unsigned, hh);
  const float r0 = opaque(x0 - __builtin_bit_cast(float, h << 16)), r1 = opaque(x1 - __builtin_bit_cast(float, h & 0xFFFF0000u));
  bf16x2 mm;
  mm[0] = (__bf16)r0; mm[1] = (__bf16)r1;
  m = __builtin_bit_cast(unsigned, mm);
  const float q0 = opaque(r0 - __builtin_bit_cast(float, m << 16)), q1 = opaque(r1 - __builtin_bit_cast(float, m & 0xFFFF0000u));
  bf16x2 ll;
  ll[0] = (__bf16)q0; ll[1] = (__bf16)q1;
  l = __builtin_bit_cast(unsigned, ll);
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

}  // namespace

// S2 (round 6, last session): the same GEMM over the 9 taps of a 3x3 stride-2 pad-1 convolution (Downsample, unet.py:58-64: a bare conv,
// no norm in front) -- k-step `it` = (32-channel chunk, tap) in the order igemm_split_weights lays the fragments out (it = chunk * 9 + tap),
// the A row of output pixel (b, oh, ow) for tap (ky, kx) is the NHWC row of input pixel (2 oh + ky - 1, 2 ow + kx - 1): a per-step uniform
// offset added to the row's base address.  Padding touches only the top row / left column of the map (even H, W): those rows load the
// centre pixel instead (an address select, no branch) and are zeroed where the staged value is picked up.
// WM = 2 (last session of round 6): the 64-column tile for Cout % 128 != 0 (the res_convs of the 128 x 128 level, Downsample 64 -> 64, data
// gradients towards 64 / 192 channels) -- the four waves sit 2 (M) x 2 (N): a wave owns ONE 32-column block and the MI m blocks of its half of
// the rows; the workgroup stages MQ = MI * WM m blocks per k-step (the staging arithmetic per MFMA doubles: these layers are bound by their A
// rows from HBM, not by the matrix pipe).
template <int MI, bool ACT, bool S2 = false, int WM = 1>
__global__ __launch_bounds__(256, 2) void k_gemm1x1_split(const ConvParams p) {
  static_assert(!(ACT && S2), "the stride-2 form has no GroupNorm affine");
  static_assert(WM == 1 || WM == 2, "waves along M");
  constexpr int NWN = 4 / WM;                   // waves side by side along N
  constexpr int MQ = MI * WM;                   // m blocks the workgroup stages per k-step
  constexpr int BM = 32 * MQ;
  constexpr int FRAG = 64 * 16;                 // bytes of one operand fragment (64 lanes x 8 bf16)
  constexpr int STAGE = 2 * MQ * 3 * FRAG;      // [K = 16 step 2][m block MQ][plane 3][FRAG]
  extern __shared__ f32x4 smem_v[];
  char* smem = reinterpret_cast<char*>(smem_v);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % NWN, wm = wave / NWN;
  const int Cin = p.C0 + p.C1;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.B * HoWo;
  const int total = (Cin >> 5) * (S2 ? 9 : 1);
  const int per = (total + p.ksplit - 1) / p.ksplit;
  const int it0 = blockIdx.y * per;
  const int it1 = min(total, it0 + per);
  const int nsteps = it1 - it0;
  const int tiles_n = p.Cout / (32 * NWN);
  // XCD-aware order: consecutive tile ids (n fastest: they share their A rows) run on ONE XCD, i.e. behind one L2
  int tile_m, tile_n;
  {
    const int nwg = gridDim.x, w = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = w & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
    tile_m = lin / tiles_n;
    tile_n = lin - tile_m * tiles_n;
  }

  // ---- loaders ---------------------------------------------------------------------------------
  const int kq = tid & 7, lrow = tid >> 3;
  int aoff0[MQ], aoff1[MQ], ssoff[MQ];
  int edge[MQ];          // S2: bit 0 = output row 0 (tap row ky = 0 is padding), bit 1 = output column 0 (kx = 0 is padding)
#pragma unroll
  for (int i = 0; i < MQ; ++i) {
    const int m = tile_m * BM + lrow + 32 * i;
    if constexpr (S2) {
      const int b = m / HoWo, r = m - b * HoWo;
      const int oh = r / p.Wo, ow = r - oh * p.Wo;
      aoff0[i] = ((b * p.Hs + 2 * oh) * p.Ws + 2 * ow) * p.C0 + kq * 4;      // the centre tap's pixel
      aoff1[i] = 0;
      edge[i] = (oh == 0 ? 1 : 0) | (ow == 0 ? 2 : 0);
    } else {
      aoff0[i] = m * p.C0 + kq * 4;
      aoff1[i] = m * p.C1 + kq * 4 - p.C0;       // + c gives the offset inside src1 for c >= C0
      edge[i] = 0;
    }
    ssoff[i] = ((m / HoWo) * Cin + kq * 4) * 2;
  }
  const bf16x8* bq = reinterpret_cast<const bf16x8*>(p.w_split) + ((size_t)(tile_n * NWN + wn) * total + it0) * (6 * 64) + lane;

  f32x4 ra[3][MQ];
  bf16x8 rb[3][6];
  f32x4 ssa[MQ], ssb[MQ];
  int pad[3] = {0, 0, 0};       // S2: bit i = row i of register set s is padding (its staged value is zero)
  auto load_a = [&](int s, int it) {
    if constexpr (S2) {
      const int chunk = it / 9, tap = it - chunk * 9;        // wave-uniform
      const int ky = tap / 3, kx = tap - ky * 3;
      const int off = ((ky - 1) * p.Ws + (kx - 1)) * p.C0 + (chunk << 5);
      const int tapedge = (ky == 0 ? 1 : 0) | (kx == 0 ? 2 : 0);
      int pm = 0;
#pragma unroll
      for (int i = 0; i < MQ; ++i) {
        const bool out = (edge[i] & tapedge) != 0;
        ra[s][i] = *reinterpret_cast<const f32x4*>(p.src0 + aoff0[i] + (out ? (chunk << 5) : off));
        pm |= out ? (1 << i) : 0;
      }
      pad[s] = pm;
      return;
    }
    const int c = it << 5;
    const bool second = c >= p.C0;          // wave-uniform
    const float* sp = second ? p.src1 : p.src0;
#pragma unroll
    for (int i = 0; i < MQ; ++i) ra[s][i] = *reinterpret_cast<const f32x4*>(sp + (second ? aoff1[i] : aoff0[i]) + c);
  };
  auto load_b = [&](int s, int rel) {
    const bf16x8* q = bq + (size_t)rel * (6 * 64);
#pragma unroll
    for (int j = 0; j < 6; ++j) rb[s][j] = q[j * 64];
  };
  auto load_ss = [&](int it) {
    if constexpr (ACT) {
      const int c2 = it << 6;
#pragma unroll
      for (int i = 0; i < MQ; ++i) {
        const float* q = p.ss + ssoff[i] + c2;
        ssa[i] = *reinterpret_cast<const f32x4*>(q);
        ssb[i] = *reinterpret_cast<const f32x4*>(q + 4);
      }
    }
  };
  // LDS write position of this thread's quad: K = 16 step kq >> 2, lane slot lrow + 32 * ((kq >> 1) & 1), 8-byte half kq & 1
  const int wbase = (kq >> 2) * (MQ * 3 * FRAG) + (lrow + 32 * ((kq >> 1) & 1)) * 16 + (kq & 1) * 8;
  auto stage_a = [&](int s, int stage) {
    char* A = smem + stage * STAGE + wbase;
#pragma unroll
    for (int i = 0; i < MQ; ++i) {
      f32x4 v = ra[s][i];
      if constexpr (S2) { if (pad[s] & (1 << i)) v = f32x4{0.f, 0.f, 0.f, 0.f}; }
      if constexpr (ACT) {
        v.x = opaque(fmaf(v.x, ssa[i].x, ssa[i].y));
        v.y = opaque(fmaf(v.y, ssa[i].z, ssa[i].w));
        v.z = opaque(fmaf(v.z, ssb[i].x, ssb[i].y));
        v.w = opaque(fmaf(v.w, ssb[i].z, ssb[i].w));
      }
      unsigned h0, m0, l0, h1, m1, l1;
      split3_plain(v.x, v.y, h0, m0, l0);
      split3_plain(v.z, v.w, h1, m1, l1);
      const u32x2 h = {h0, h1}, m = {m0, m1}, l = {l0, l1};
      *reinterpret_cast<u32x2*>(A + (i * 3 + 0) * FRAG) = h;
      *reinterpret_cast<u32x2*>(A + (i * 3 + 1) * FRAG) = m;
      *reinterpret_cast<u32x2*>(A + (i * 3 + 2) * FRAG) = l;
    }
  };

  f32x16 acc[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // step i (register set i % 3, LDS stage i & 1): the LDS holds step i; set (i + 1) % 3 holds step i + 1 (loaded one step ago) and is
  // staged into the other LDS stage BETWEEN this step's MFMAs; set (i + 2) % 3 is loaded now.  Branch-free (the last two steps re-load /
  // re-stage the last k-step into registers and an LDS stage nobody reads again), and hand-placed: the staging arithmetic is cut into
  // slices of <= 4 plain VALU instructions, a slice behind an MFMA, each (MFMA, slice) group pinned with sched_barrier(0) -- left to
  // itself (and under sched_group_barrier) the compiler clusters the MFMAs and the VALU work, and a cluster of VALU hides under nothing
  // (profiles/r05a_mfma_fillers.txt: <= 5 plain VALU per MFMA are free when interleaved, clustered they are paid in full).
  // The step's ONE barrier sits two thirds into its MFMAs: the staging is complete by then, and behind the barrier the wave already
  // reads the NEXT step's first fragments -- the LDS round trip and the barrier skew run under the last third of the MFMAs instead of
  // in front of the next step's first one.
  // Slices per staged quad (4 floats): affine | per pair: h, first residuals, m, second residuals + l | the three LDS writes.
  constexpr int NSQ = 10, NSL = NSQ * MQ, NMF = 12 * MI, NBAR = 8 * MI;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // smallest terms first (mfma_split6's order)
  bf16x8 a0[MI][3];
  auto read_a0 = [&](int stage) {
    const char* Ar = smem + stage * STAGE + lane * 16 + wm * (MI * 3 * FRAG);      // this wave's m blocks: wm * MI ...
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) a0[m][pl] = *reinterpret_cast<const bf16x8*>(Ar + (m * 3 + pl) * FRAG);
  };
  auto step = [&](int s0, int s1, int s2, int i) {
    const int stage = i & 1;
    const int i2 = min(i + 2, nsteps - 1);
    const char* Ar = smem + stage * STAGE + lane * 16 + wm * (MI * 3 * FRAG);
    char* Aw = smem + (stage ^ 1) * STAGE + wbase;
    bf16x8 a1[MI][3];
#ifdef SR3_G1_ABL      // timing-only A/B builds (tools/build_variant.sh): 1 = every step re-reads B of step 0 (L1-hot), 2 = no staging
                       // arithmetic / LDS writes, 4 = every step re-reads A of step 0, 8 = no MFMAs
    load_a(s2, (SR3_G1_ABL & 4) ? it0 : it0 + i2); load_b(s2, (SR3_G1_ABL & 1) ? 0 : i2);
#else
    load_a(s2, it0 + i2); load_b(s2, i2);
#endif
    __builtin_amdgcn_sched_barrier(0);
    f32x4 v[MQ];
    float r[MQ][4];
    unsigned hh[MQ][2], mm[MQ][2], ll[MQ][2];
    auto slice = [&](int sl) {
#ifdef SR3_G1_ABL
      if (SR3_G1_ABL & 2) return;
#endif
      const int j = sl / NSQ, ph = sl % NSQ;
      if (ph == 0) {
        v[j] = ra[s1][j];
        if constexpr (S2) { if (pad[s1] & (1 << j)) v[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if constexpr (ACT) {
          v[j].x = opaque(fmaf(v[j].x, ssa[j].x, ssa[j].y));
          v[j].y = opaque(fmaf(v[j].y, ssa[j].z, ssa[j].w));
          v[j].z = opaque(fmaf(v[j].z, ssb[j].x, ssb[j].y));
          v[j].w = opaque(fmaf(v[j].w, ssb[j].z, ssb[j].w));
        }
      } else if (ph == 9) {                      // the quad's three planes
        const u32x2 h = {hh[j][0], hh[j][1]}, m = {mm[j][0], mm[j][1]}, l = {ll[j][0], ll[j][1]};
        *reinterpret_cast<u32x2*>(Aw + (j * 3 + 0) * FRAG) = h;
        *reinterpret_cast<u32x2*>(Aw + (j * 3 + 1) * FRAG) = m;
        *reinterpret_cast<u32x2*>(Aw + (j * 3 + 2) * FRAG) = l;
      } else {
        const int e = (ph - 1) >> 2, sub = (ph - 1) & 3;
        if (sub == 0) {                          // h of the pair
          bf16x2 t;
          t[0] = (__bf16)v[j][2 * e]; t[1] = (__bf16)v[j][2 * e + 1];
          hh[j][e] = __builtin_bit_cast(unsigned, t);
        } else if (sub == 1) {                   // first residuals
          r[j][2 * e] = opaque(v[j][2 * e] - __builtin_bit_cast(float, hh[j][e] << 16));
          r[j][2 * e + 1] = opaque(v[j][2 * e + 1] - __builtin_bit_cast(float, hh[j][e] & 0xFFFF0000u));
        } else if (sub == 2) {                   // m
          bf16x2 t;
          t[0] = (__bf16)r[j][2 * e]; t[1] = (__bf16)r[j][2 * e + 1];
          mm[j][e] = __builtin_bit_cast(unsigned, t);
        } else {                                 // second residuals, l
          const float q0 = opaque(r[j][2 * e] - __builtin_bit_cast(float, mm[j][e] << 16));
          const float q1 = opaque(r[j][2 * e + 1] - __builtin_bit_cast(float, mm[j][e] & 0xFFFF0000u));
          bf16x2 t;
          t[0] = (__bf16)q0; t[1] = (__bf16)q1;
          ll[j][e] = __builtin_bit_cast(unsigned, t);
        }
      }
    };
#pragma unroll
    for (int n = 0; n < NMF; ++n) {
      const int ks = n / (6 * MI), q = (n % (6 * MI)) / MI, m = n % MI;
#ifdef SR3_G1_ABL
      if (SR3_G1_ABL & 8) { if (q == 0) { acc[m][0] += (float)a0[m][0][0] + (float)a1[m][0][0] + (float)rb[s0][0][0] + (float)rb[s0][3][0]; } } else
#endif
      if (ks == 0) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[m][PA[q]], rb[s0][PB[q]], acc[m], 0, 0, 0);
      else acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[m][PA[q]], rb[s0][3 + PB[q]], acc[m], 0, 0, 0);
      if (n == 2 * MI) {                         // the second K = 16 step's fragments, 4 MI MFMAs ahead of their use
#pragma unroll
        for (int mb = 0; mb < MI; ++mb)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) a1[mb][pl] = *reinterpret_cast<const bf16x8*>(Ar + ((MQ + mb) * 3 + pl) * FRAG);
      }
      if (n < NBAR - 1) {
#pragma unroll
        for (int sl = (n * NSL) / (NBAR - 1); sl < ((n + 1) * NSL) / (NBAR - 1); ++sl) slice(sl);
      }
      if (n == NBAR - 1) {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                         // (s_waitcnt lgkmcnt(0) + s_barrier: the staged rows are visible, this stage's reads are done)
        read_a0(stage ^ 1);
        load_ss(it0 + i2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (nsteps > 0) {
    load_a(0, it0); load_b(0, 0); load_ss(it0);
    if (nsteps > 1) { load_a(1, it0 + 1); load_b(1, 1); }
    stage_a(0, 0);
    if (nsteps > 1) load_ss(it0 + 1);
    __syncthreads();
    read_a0(0);
    int i = 0;
    for (; i + 2 < nsteps; i += 3) {
      step(0, 1, 2, i);
      step(1, 2, 0, i + 1);
      step(2, 0, 1, i + 2);
    }
    if (i < nsteps) step(0, 1, 2, i);
    if (i + 1 < nsteps) step(1, 2, 0, i + 1);
  }

  // ---- epilogue: D layout of the 32x32 MFMA: reg r of lane l -> row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col l & 31 ----------
  const bool direct = p.ksplit == 1;
  float* dst = direct ? p.out : p.partial + (size_t)blockIdx.y * M * p.Cout;
  const int n = tile_n * (32 * NWN) + wn * 32 + (lane & 31);
  float bn = 0.f;
  if (direct && p.bias) bn = p.bias[n];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int mbase = tile_m * BM + 32 * (wm * MI + i) + 8 * g;       // wave-uniform; an 8-row group never straddles an image (HoWo % 32 == 0)
      float fb = bn;
      if (direct && p.film) fb += p.film[(size_t)(mbase / HoWo) * p.film_stride + n];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = mbase + e + 4 * (lane >> 5);
        float v = acc[i][g * 4 + e];
        if (direct) {
          v += fb;
          if (p.res0) v += (n < p.RC0) ? p.res0[(size_t)m * p.RC0 + n] : p.res1[(size_t)m * p.RC1 + (n - p.RC0)];
        }
        dst[(size_t)m * p.Cout + n] = v;
      }
    }
  }
}

bool gemm1x1_s2(const ConvParams& p) {      // the 3x3 stride-2 pad-1 form (Downsample): one source, no activation, even map
  return p.ksize == 3 && p.stride == 2 && p.C1 == 0 && p.act == 0 && !(p.Hs & 1) && !(p.Ws & 1) && p.Ho * 2 == p.Hs && p.Wo * 2 == p.Ws;
}
bool gemm1x1_fits(const ConvParams& p, int mi) {
  const long M = (long)p.B * p.Ho * p.Wo;
  return ((p.ksize == 1 && p.stride == 1) || gemm1x1_s2(p)) && p.ups == 0 && (p.Cout & 63) == 0 && p.C0 > 0 && (p.C0 & 31) == 0 && (p.C1 & 31) == 0 &&
         M % (32 * mi) == 0 && ((p.Ho * p.Wo) & 31) == 0 && (p.act == 0 || p.act == 1) && p.drop_thresh == 0 && !p.x2_w;
}

// rows of the tile: 64 (two m blocks per wave), or 32 where 64 would leave workgroup slots empty -- 512 slots = two workgroups per CU; the
// N = 512 layers at 16 x 16 (M = 4096) give 256 tiles of 64 rows: one wave per SIMD, nothing to overlap with; 512 tiles of 32 rows fill them
int gemm1x1_cols(const ConvParams& p) { return (p.Cout & 127) ? 64 : 128; }      // the 2 x 2 wave arrangement (WM = 2) where Cout % 128 != 0
int gemm1x1_rows(const ConvParams& p) {
  if (p.Cout & 127) return 64;
#ifdef SR3_G1_NO32
  return 64;
#elif defined(SR3_G1_ALL32)
  return 32;
#else
  // measured in the C2 forward (profiles/r06_gemm1x1.txt, item 7): 512 -> 512 at 16 x 16 29 -> 25.6 us and the 8 x 8 maps' layers
  // without split-K (18.6 / 22.3 us against 21.2 / 25.5 with it) gain; 1024 -> 512 and 768 -> 512 at 16 x 16 lose 1-2 us (long K: the
  // 64-row tile's reuse of the weight fragments wins), so they keep 64 rows
  const long M = (long)p.B * p.Ho * p.Wo;
  if ((M / 64) * (p.Cout / 128) >= 384) return 64;
  if (p.ksize == 3) return 32;       // (stride-2 form: long K, few rows -- 32-row tiles and split-K fill the slots, conv_pick)
  return (M <= 1024 || p.C0 + p.C1 <= 512) ? 32 : 64;
#endif
}

int gemm1x1_forward(const ConvParams& p, int mi, hipStream_t st) {
  if (!gemm1x1_fits(p, mi) || !p.w_split) { set_error("conv: the 1x1 GEMM kernel (tile 22) does not fit this problem"); return SR3_E_UNSUPPORTED; }
  if (p.act == 1 && !p.ss) { set_error("conv: act needs ss"); return SR3_E_BADARG; }
  const int M = p.B * p.Ho * p.Wo;
  const int rows = gemm1x1_rows(p), cols = gemm1x1_cols(p);
  mi = rows / 32;
  dim3 grid((M / rows) * (p.Cout / cols), p.ksplit);
  const int smem = 2 * 2 * mi * 3 * 1024;
  static std::atomic<uint64_t> done[9];
  auto go = [&](auto kern, std::atomic<uint64_t>& d) -> int {
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), smem, d)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
    SR3_LAUNCH_CHECK("k_gemm1x1_split");
    return SR3_OK;
  };
  // (MI = 4, a 128-row tile with half the weight traffic per row, was built and dropped: at 256 registers it spills, and the weight
  // traffic is not what bounds this kernel -- profiles/r06_gemm1x1.txt)
  if (p.ksize == 3 && p.act) { set_error("conv: the stride-2 form of the GEMM kernel takes no activation"); return SR3_E_UNSUPPORTED; }
  if (cols == 64) {      // 64 x 64 tile, waves 2 x 2 (one m block per wave)
    if (p.ksize == 3) return go(k_gemm1x1_split<1, false, true, 2>, done[6]);
    return p.act ? go(k_gemm1x1_split<1, true, false, 2>, done[7]) : go(k_gemm1x1_split<1, false, false, 2>, done[8]);
  }
  if (p.ksize == 3) {
    return mi == 2 ? go(k_gemm1x1_split<2, false, true>, done[4]) : go(k_gemm1x1_split<1, false, true>, done[5]);
  }
  if (mi == 2) return p.act ? go(k_gemm1x1_split<2, true>, done[0]) : go(k_gemm1x1_split<2, false>, done[1]);
  if (mi == 1) return p.act ? go(k_gemm1x1_split<1, true>, done[2]) : go(k_gemm1x1_split<1, false>, done[3]);
  set_error("conv: bad 1x1 GEMM tile");
  return SR3_E_BADARG;
}

}  // namespace sr3
