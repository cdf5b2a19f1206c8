// 3x3 stride-1 convolution (the 92 %-of-FLOPs op of the SR3 UNet: `Block` = GroupNorm -> Swish ->
// Conv3x3, model/sr3_modules/unet.py:80-91, plus Upsample's conv, :58-65) as an implicit GEMM on
// v_mfma_f32_32x32x2_f32, organised around an LDS-resident *activated halo tile*.
//
// Why not plain im2col staging (conv_igemm.hip): on gfx950 the exact-fp32 MFMA runs at the fp32
// VALU rate and, measured with rocprofv3 (profiles/archive/r01_*), VALU work does not overlap it -- every
// VALU instruction spent on staging is MFMA time lost.  With im2col staging each input element is
// loaded, GroupNorm-scaled and SiLU'd once per filter tap (9x); here a workgroup owns a spatial
// output tile (TH x TW pixels of NB images), stages the (TH+2) x (TW+2) input halo of one 32-channel
// chunk ONCE (gather for the x2 nearest upsample and the skip-concat seam included, GN+SiLU applied
// once, zero padding applied after the activation), and the 9 taps read their A fragments from that
// tile with shifted addresses.  Only the weights of the current tap are re-staged per k-step
// (double-buffered).  VALU per MFMA drops ~6x.
//
// GEMM view per workgroup: out[m][n] += A_tap[m][k] * W_tap[n][k], m = pixel of the tile,
// k = 32 channels of the chunk.  Waves: WAVES_M x WAVES_N, each a 64 x 64 (MI = NI = 2) block of
// 32x32 MFMA tiles.  Fragment k-ordering as in conv_igemm.hip (one ds_read_b128 = 4 k-steps).
// Epilogue: accumulators are transposed through wave-private LDS so that bias / FiLM / residual
// loads and the NHWC stores are 16-byte wide; optional fused per-(image, channel) statistics of the
// output for the next GroupNorm: one partial {sum, sumsq} (double) per (image, tile, channel), written with
// plain stores and summed in a fixed order by the fold kernel (no atomics, bitwise reproducible).
//
// Instantiations: 4 waves (128x128 or 256x64 tile, two workgroups per CU) or 8 waves (256x128, one per CU);
// X2 = a fused second K-segment (the 1x1 res_conv); DROP = train-mode dropout in the staging step; MODE 1 =
// opt-in split-bf16 MFMA (see below).  Workgroup order is XCD-aware.
#include <stdlib.h>

#include <type_traits>

#include "sr3_common.h"

namespace sr3 {

__device__ __forceinline__ float silu_h(float v) { return SR3_SILU(v); }


// MODE 0: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).  MODE 1 (opt-in, experimental): every fp32 operand is
// split into three bf16 terms in the staging step and each product is evaluated as the six bf16 MFMA products
// hh + hm + mh + mm + hl + lh (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the dropped terms are <= 2^-23 of
// the product, i.e. fp32-class accuracy at 6/16 of the matrix-pipe time of the fp32 instruction.
template <int WAVES_M, int WAVES_N, bool X2, bool DROP, int MODE, int NI>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, WAVES_M * WAVES_N == 4 ? 2 : 1)
void k_conv3x3_halo(const ConvParams p, const HaloGeom g) {
  constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;     // 4 waves (two workgroups per CU) or 8 (one)
  constexpr int RPP = NT / 8;                 // tile rows one loader pass covers (8 threads x float4 per row)
  constexpr int LDK = 36, BK = 32;
  constexpr int LDB = 32;                     // MODE 1: bf16 row stride, 64 bytes, no padding: the four 16-byte segments of a
                                              // row are XOR-swizzled with (row >> 2) & 3, which makes both the 8-byte staging
                                              // writes (4 rows x 64 B per 32 lanes) and the 16-byte fragment reads (16
                                              // consecutive rows x 16 B) bank-conflict free
  auto swz = [](int row, int seg) { return row * LDB + (((seg ^ (row >> 2)) & 3) << 3); };   // bf16 index of a segment
  constexpr int WST = (MODE == 1 && NW == 4) ? 1 : 2;   // weight stages: the 4-wave MODE 1 tile single-buffers to keep two workgroups per CU
  constexpr int WCOLS = 32 * NI;              // columns of a wave's tile: NI 32x32 MFMA tiles side by side (64 rows x WCOLS)
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * WCOLS;
  constexpr int BR = BN / RPP;                // weight loader rows per thread
  constexpr int HP_MAX = (BM == 128) ? 200 : 324;
  constexpr int HI = (HP_MAX * 8 + NT - 1) / NT; // halo float4 items per thread
  constexpr int WSTAGE = BN * LDK;
  extern __shared__ f32x4 smem_v[];
  float* smem = reinterpret_cast<float*>(smem_v);
  float* halo = smem;                         // MODE 0: [HP][LDK]
  float* wst = smem + HP_MAX * LDK;           // MODE 0: 2 x [BN][LDK]
  __bf16* halo_b = reinterpret_cast<__bf16*>(smem_v);         // MODE 1: 3 planes x [HP_MAX][LDB]
  __bf16* wst_b = halo_b + 3 * HP_MAX * LDB;                  // MODE 1: 3 planes x [BN][LDB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = tid & 7, lrow = tid >> 3;
  const int Cin = p.C0 + p.C1;
  const int H = p.Ho, W = p.Wo;               // stride 1: output dims == virtual input dims
  const int tiles_n = (p.Cout + BN - 1) / BN;
  int bid = blockIdx.x;
  // XCD-aware order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs (each with its own L2), so
  // give every XCD one contiguous range of tiles -- the Cout siblings of a spatial tile and its neighbours then share
  // their input halo through a single L2 instead of fetching it once per XCD.
  if ((gridDim.x & 7) == 0 && !(p.dbg & 4)) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const int tile_n = bid % tiles_n;
  bid /= tiles_n;
  const int tw_i = bid % g.tiles_w;
  bid /= g.tiles_w;
  const int th_i = bid % g.tiles_h;
  const int tb_i = bid / g.tiles_h;           // image-group index
  const int h0 = th_i * g.TH, w0 = tw_i * g.TW, b0 = tb_i * g.NB;

  // Segment 1 (the 3x3 conv): 32-channel chunks x 9 taps, split-K over chunks.  Segment 2 (X2: the
  // fused 1x1 conv of x2, centre tap only) runs as a second, separate loop in the LAST split, so the
  // main loop carries no per-segment selects.
  const int Cin2 = X2 ? p.x2_C0 + p.x2_C1 : 0;
  const int nch1 = (Cin + BK - 1) / BK;
  const int nch2 = (Cin2 + BK - 1) / BK;
  const int cper = (nch1 + p.ksplit - 1) / p.ksplit;
  const int c_begin = blockIdx.y * cper;
  const int c_end = min(nch1, c_begin + cper);
  const bool do_x2 = X2 && (int)blockIdx.y == p.ksplit - 1;

  // ---- halo items of this thread (fixed for the whole kernel) --------------------------------
  // item j covers halo pixel (tid >> 3) + RPP j, channel quad kq of the current chunk
  int hpix[HI];        // source pixel index ((b*Hs + y)*Ws + x), or -1 when the tap falls in the padding
  int himg[HI];        // image slot within the tile (for the GN scale/shift select)
  const int TWp = g.TW + 2;
#pragma unroll
  for (int j = 0; j < HI; ++j) {
    const int hp = lrow + RPP * j;
    int pix = -1, nb = 0;
    if (hp < g.HP) {
      nb = hp / g.HPI;
      const int r = hp - nb * g.HPI;
      const int hy = r / TWp, hx = r - hy * TWp;
      const int ih = h0 + hy - 1, iw = w0 + hx - 1;
      const int b = b0 + nb;
      if (b < p.B && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
        pix = (b * p.Hs + (ih >> p.ups)) * p.Ws + (iw >> p.ups);
    }
    hpix[j] = pix;
    himg[j] = nb;
  }

  // weight tiles in flight: the 8-wave split-bf16 kernel prefetches two taps ahead (its taps are ~3x shorter than the
  // fp32 kernel's, about one L2 round trip), every other instantiation one
  constexpr bool PF2 = MODE == 1 && NW == 8;
  constexpr int WSETS = PF2 ? 2 : 1;
  using SET0 = std::integral_constant<int, 0>;
  using SET1 = std::integral_constant<int, WSETS - 1>;
  f32x4 rh[HI], rw[WSETS][BR];
  f32x4 ssa[2], ssb[2];     // scale/shift of this thread's channel quad for image slots 0 / 1
  bool wok[WSETS][BR];
  bool hvalid = false;      // channel quad of the staged chunk is inside Cin

  int cur_act = 0;          // prologue of the staged chunk (segment 2 has none)
  int cur_c = 0;            // first channel of this thread's quad in the staged chunk
  auto load_halo = [&](auto seg2_tag, int chunk) {
    constexpr bool seg2 = decltype(seg2_tag)::value;
    const int CinS = seg2 ? Cin2 : Cin;
    const int C0S = seg2 ? p.x2_C0 : p.C0;
    const int c = chunk * BK + kq * 4;
    hvalid = c < CinS;
    cur_c = c;
    cur_act = seg2 ? 0 : p.act;
    const int ce = hvalid ? c : 0;
    const bool second = ce >= C0S;
    const float* sp = seg2 ? (second ? p.x2_src1 : p.x2_src0) : (second ? p.src1 : p.src0);
    const int sC = seg2 ? (second ? p.x2_C1 : p.x2_C0) : (second ? p.C1 : p.C0);
    const int cs = second ? ce - C0S : ce;
#pragma unroll
    for (int j = 0; j < HI; ++j) {
      const int off = hpix[j] >= 0 ? hpix[j] * sC + cs : 0;
      rh[j] = *reinterpret_cast<const f32x4*>(sp + off);
    }
    if (cur_act != 0) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int b = min(b0 + s, p.B - 1);
        const float* q = p.ss + (b * Cin + ce) * 2;
        ssa[s] = *reinterpret_cast<const f32x4*>(q);
        ssb[s] = *reinterpret_cast<const f32x4*>(q + 4);
      }
    }
  };

  auto store_halo = [&]() {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < HI; ++j) {
      const int hp = lrow + RPP * j;
      if (hp < g.HP) {
        f32x4 v = rh[j];
        if (cur_act != 0) {
          const f32x4 sa = himg[j] ? ssa[1] : ssa[0];
          const f32x4 sb = himg[j] ? ssb[1] : ssb[0];
          v.x = fmaf(v.x, sa.x, sa.y);
          v.y = fmaf(v.y, sa.z, sa.w);
          v.z = fmaf(v.z, sb.x, sb.y);
          v.w = fmaf(v.w, sb.z, sb.w);
          if (cur_act == 2) { v.x = silu_h(v.x); v.y = silu_h(v.y); v.z = silu_h(v.z); v.w = silu_h(v.w); }
          if (DROP) {                    // segment 1 only (cur_act != 0), single source => linear NHWC index
            const unsigned i0 = (unsigned)(hpix[j] * p.C0 + cur_c);
            v.x *= drop_mask(p.drop_seed, i0, p.drop_thresh, p.drop_scale);
            v.y *= drop_mask(p.drop_seed, i0 + 1, p.drop_thresh, p.drop_scale);
            v.z *= drop_mask(p.drop_seed, i0 + 2, p.drop_thresh, p.drop_scale);
            v.w *= drop_mask(p.drop_seed, i0 + 3, p.drop_thresh, p.drop_scale);
          }
        }
        v = (hvalid && hpix[j] >= 0) ? v : zero;
        if constexpr (MODE == 0) {
          *reinterpret_cast<f32x4*>(&halo[hp * LDK + kq * 4]) = v;
        } else {
          bf16x4 h, m, l;
          split3(v, h, m, l);
          const int o = swz(hp, kq >> 1) + (kq & 1) * 4;
          *reinterpret_cast<bf16x4*>(&halo_b[0 * HP_MAX * LDB + o]) = h;
          *reinterpret_cast<bf16x4*>(&halo_b[1 * HP_MAX * LDB + o]) = m;
          *reinterpret_cast<bf16x4*>(&halo_b[2 * HP_MAX * LDB + o]) = l;
        }
      }
    }
  };

  auto load_w = [&](auto seg2_tag, auto set_tag, int chunk, int tap) {
    constexpr bool seg2 = decltype(seg2_tag)::value;
    constexpr int S = decltype(set_tag)::value;
    const int CinS = seg2 ? Cin2 : Cin;
    const float* wp = seg2 ? p.x2_w : p.w;
    constexpr int ntap = seg2 ? 1 : 9;
    const int c = chunk * BK + kq * 4;
    const bool cvalid = c < CinS;
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      const int n = tile_n * BN + lrow + RPP * j;
      const bool ok = cvalid && n < p.Cout;
      wok[S][j] = ok;
      const int off = ok ? (n * ntap + tap) * CinS + c : 0;
      rw[S][j] = *reinterpret_cast<const f32x4*>(wp + off);
    }
  };
  auto store_w = [&](auto set_tag, int stage) {
    constexpr int S = decltype(set_tag)::value;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == 0) {
      float* Bw = wst + stage * WSTAGE;
#pragma unroll
      for (int j = 0; j < BR; ++j)
        *reinterpret_cast<f32x4*>(&Bw[(lrow + RPP * j) * LDK + kq * 4]) = wok[S][j] ? rw[S][j] : zero;
    } else {
      __bf16* Bb = wst_b + stage * 3 * BN * LDB;
#pragma unroll
      for (int j = 0; j < BR; ++j) {
        bf16x4 h, m, l;
        split3(wok[S][j] ? rw[S][j] : zero, h, m, l);
        const int o = swz(lrow + RPP * j, kq >> 1) + (kq & 1) * 4;
        *reinterpret_cast<bf16x4*>(&Bb[0 * BN * LDB + o]) = h;
        *reinterpret_cast<bf16x4*>(&Bb[1 * BN * LDB + o]) = m;
        *reinterpret_cast<bf16x4*>(&Bb[2 * BN * LDB + o]) = l;
      }
    }
  };

  // ---- MFMA fragments ---------------------------------------------------------------------------
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int kh = (lane >> 5) * 4;
  int hbase[2];     // halo pixel of this lane's A row for tap (0,0), per 32-row block
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = wave_m * 64 + i * 32 + (lane & 31);
    const int nb = m >> g.log_thw;
    const int ty = (m >> g.log_tw) & (g.TH - 1);
    const int tx = m & (g.TW - 1);
    hbase[i] = nb * g.HPI + ty * TWp + tx;
  }
  const int brow = wave_n * WCOLS + (lane & 31);

  f32x16 acc[2][NI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int stage, int shift) {
    if constexpr (MODE == 1) {
      const __bf16* Bb = wst_b + stage * 3 * BN * LDB;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {          // two K = 16 steps per 32-channel chunk
        const int seg = ks * 2 + (lane >> 5);       // 16-byte segment (8 channels) of the 32-channel row
        bf16x8 a[2][3], b[NI][3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int o = swz(hbase[i] + shift, seg);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) a[i][pl] = *reinterpret_cast<const bf16x8*>(&halo_b[pl * HP_MAX * LDB + o]);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int o = swz(brow + 32 * j, seg);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) b[j][pl] = *reinterpret_cast<const bf16x8*>(&Bb[pl * BN * LDB + o]);
        }
        // product-major order: four independent accumulators between dependent MFMAs; smallest terms first
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
      }
      return;
    }
    const float* Bw = wst + stage * WSTAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(&halo[(hbase[i] + shift) * LDK + kk * 8 + kh]);
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const f32x4*>(&Bw[(brow + 32 * j) * LDK + kk * 8 + kh]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
    }
  };

  // ---- main loop: chunks (halo restage) x 9 taps (weight restage) ---------------------------------
  using SEG1 = std::integral_constant<bool, false>;
  using SEG2 = std::integral_constant<bool, true>;
  if constexpr (PF2) {
    // Weight tiles are prefetched two taps ahead; tile tl (linear tap index over chunks x 9) lives in register set
    // tl & 1, so the loop is unrolled by two to keep the set a compile-time constant.
    const int ntl = (c_end - c_begin) * 9;
    auto load_tile = [&](auto set_tag, int tl) {
      const int ch = tl / 9;
      load_w(SEG1{}, set_tag, c_begin + ch, tl - ch * 9);
    };
    if (ntl > 0) {
      load_halo(SEG1{}, c_begin);
      load_tile(SET0{}, 0);
      store_halo();
      store_w(SET0{}, 0);
      if (ntl > 1) load_tile(SET1{}, 1);
      __syncthreads();
      int stage = 0;
      auto body = [&](auto next_set /* holds tile tl + 1 */, auto free_set /* held tile tl, already staged */, int tl) {
        const int ch = tl / 9, tap = tl - ch * 9;
        const int chunk = c_begin + ch;
        const bool more_chunks = chunk + 1 < c_end;
        const bool last_tap = tap == 8;
        if (tap == 0 && more_chunks) load_halo(SEG1{}, chunk + 1);     // in flight across the 9 taps of this chunk
        if (tl + 2 < ntl) load_tile(free_set, tl + 2);
        const int fr = tap / 3, fs = tap - fr * 3;
        compute(stage, fr * TWp + fs);
        if (tl + 1 < ntl) store_w(next_set, stage ^ 1);
        if (last_tap && more_chunks) {
          __syncthreads();                          // every wave is done reading the halo tile
          store_halo();
        }
        __syncthreads();
        stage ^= 1;
      };
#pragma unroll 1
      for (int tl = 0; tl < ntl; tl += 2) {
        body(SET1{}, SET0{}, tl);
        if (tl + 1 < ntl) body(SET0{}, SET1{}, tl + 1);
      }
    }
  } else if (c_begin < c_end) {
    load_halo(SEG1{}, c_begin);
    load_w(SEG1{}, SET0{}, c_begin, 0);
    store_halo();
    store_w(SET0{}, 0);
    __syncthreads();
    int stage = 0;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
      const bool more_chunks = chunk + 1 < c_end;
      if (more_chunks && !(p.dbg & 2)) load_halo(SEG1{}, chunk + 1);   // in flight across the 9 taps of this chunk
#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {
        const bool last_tap = tap == 8;
        const bool more = !last_tap || more_chunks;
        if (more && !(p.dbg & 2)) load_w(SEG1{}, SET0{}, last_tap ? chunk + 1 : chunk, last_tap ? 0 : tap + 1);
        const int fr = tap / 3, fs = tap - fr * 3;
        if (!(p.dbg & 1)) compute(stage, fr * TWp + fs);
        if constexpr (WST == 1) __syncthreads();    // single weight stage: every wave is done with it
        if (more && !(p.dbg & 2)) store_w(SET0{}, stage ^ (WST - 1));
        if (last_tap && more_chunks && !(p.dbg & 2)) {
          if constexpr (WST == 2) __syncthreads();  // every wave is done reading the halo tile
          store_halo();
        }
        __syncthreads();
        stage ^= WST - 1;
      }
    }
  }
  if constexpr (X2) {
    // ---- segment 2: 1x1 conv of x2, one k-step per chunk (halo centre, its own weights) ------------
    if (do_x2 && nch2 > 0) {
      load_halo(SEG2{}, 0);
      load_w(SEG2{}, SET0{}, 0, 0);
      store_halo();          // the last barrier of the main loop already retired every LDS read
      store_w(SET0{}, 0);
      __syncthreads();
      int stage = 0;
      for (int chunk = 0; chunk < nch2; ++chunk) {
        const bool more = chunk + 1 < nch2;
        if (more) { load_halo(SEG2{}, chunk + 1); load_w(SEG2{}, SET0{}, chunk + 1, 0); }
        compute(stage, TWp + 1);
        if (more) {
          if constexpr (WST == 1) __syncthreads();
          store_w(SET0{}, stage ^ (WST - 1));
          if constexpr (WST == 2) __syncthreads();
          store_halo();
        }
        __syncthreads();
        stage ^= WST - 1;
      }
    }
  }

  // ---- epilogue: wave-private LDS transpose, 16-byte bias / FiLM / residual / store ----------------
  // D layout: reg r of lane l -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31.
  __syncthreads();                                 // all MFMA reads of LDS are complete
  constexpr int LDT = 68;                          // 64 + 4 floats
  float* tr = smem + wave * (32 * LDT);
  double* sred = reinterpret_cast<double*>(smem + NW * 32 * LDT);   // [RB][BN][2] after the transpose regions
  const bool direct = p.ksplit == 1;
  const size_t Mtot = (size_t)p.B * H * W;
  float* dst = direct ? p.out : p.partial + (size_t)blockIdx.y * Mtot * p.Cout;
  constexpr int C4N = 8 * NI;                      // float4 columns of the wave tile; 64 / C4N rows per pass
  const int c4 = lane % C4N;                       // this lane's float4 column within the wave tile
  const int n = tile_n * BN + wave_n * WCOLS + c4 * 4;
  const bool nok = n < p.Cout;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (direct && nok && p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
  if (X2 && direct && nok && p.x2_bias) bias4 += *reinterpret_cast<const f32x4*>(p.x2_bias + n);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tr[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + 32 * j + (lane & 31)] = acc[i][j][r];
    // wave-private region: LDS executes a wave's instructions in order, so only the data return
    // has to be awaited (no workgroup barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int mblk = wave_m * 64 + i * 32;                       // first tile row of this 32-row block
    const int nb = mblk >> g.log_thw;                            // uniform: a block never straddles images
    const int b = b0 + nb;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    f32x4 film4 = {0.f, 0.f, 0.f, 0.f};
    if (direct && nok && p.film && b < p.B) film4 = *reinterpret_cast<const f32x4*>(p.film + (size_t)b * p.film_stride + n);
#pragma unroll
    for (int e = 0; e < C4N / 2; ++e) {
      const int row = lane / C4N + (64 / C4N) * e;
      const int m = mblk + row;
      const int ty = (m >> g.log_tw) & (g.TH - 1);
      const int tx = m & (g.TW - 1);
      f32x4 v = *reinterpret_cast<const f32x4*>(&tr[row * LDT + c4 * 4]);
      if (b < p.B && nok) {
        const size_t pix = ((size_t)b * H + (h0 + ty)) * W + (w0 + tx);
        if (direct) {
          v += bias4 + film4;
          if (p.res0) {
            if (n < p.RC0) v += *reinterpret_cast<const f32x4*>(p.res0 + pix * p.RC0 + n);
            else v += *reinterpret_cast<const f32x4*>(p.res1 + pix * p.RC1 + (n - p.RC0));
          }
          if (p.ostat) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const double dv = (double)v[k]; s1[k] += dv; s2[k] += dv * dv; }
          }
        }
        *reinterpret_cast<f32x4*>(dst + pix * p.Cout + n) = v;
      }
    }
    if (direct && p.ostat) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = C4N; o < 64; o <<= 1) { s1[k] += __shfl_xor(s1[k], o); s2[k] += __shfl_xor(s2[k], o); }
      }
      if (lane < C4N) {
        // sred[row block][column of the tile][2]; row block = wave_m * 2 + i
        double* o = sred + ((size_t)(wave_m * 2 + i) * BN + wave_n * WCOLS + c4 * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[2 * k] = s1[k]; o[2 * k + 1] = s2[k]; }
      }
    }
  }
  if (direct && p.ostat) {
    // workgroup-level reduction of the row blocks that belong to one image, then ONE plain store
    // per (image, tile, channel): partial statistics, summed later by the fold kernel.
    __syncthreads();
    constexpr int RB = WAVES_M * 2;                     // 32-row blocks in the tile
    const int rb_per_img = RB / g.NB;
    const int T = g.NB == 1 ? g.tiles_h * g.tiles_w : 1;
    const int tix = g.NB == 1 ? th_i * g.tiles_w + tw_i : 0;
    for (int idx = tid; idx < BN * g.NB; idx += NT) {
      const int col = idx % BN, nb = idx / BN;
      const int nn = tile_n * BN + col;
      const int b = b0 + nb;
      if (nn < p.Cout && b < p.B) {
        double a1 = 0.0, a2 = 0.0;
        for (int r = 0; r < rb_per_img; ++r) {
          const double* q = sred + ((size_t)(nb * rb_per_img + r) * BN + col) * 2;
          a1 += q[0]; a2 += q[1];
        }
        double* o = p.ostat + (((size_t)b * T + tix) * p.Cout + nn) * 2;
        o[0] = a1; o[1] = a2;
      }
    }
  }
}

// ---- host -----------------------------------------------------------------------------------------
namespace {
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

template <int WAVES_M, int WAVES_N, bool X2, bool DROP, int MODE, int NI>
int launch_halo(const ConvParams& p, const HaloGeom& g, hipStream_t st) {
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 32 * NI;
  constexpr int HP_MAX = (BM == 128) ? 200 : 324;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WST = (MODE == 1 && NW == 4) ? 1 : 2;
  constexpr int smem_main = MODE == 0 ? (HP_MAX * 36 + 2 * BN * 36) * 4 : (3 * HP_MAX * 32 + WST * 3 * BN * 32) * 2;
  constexpr int smem_epi = NW * 32 * 68 * 4 + WAVES_M * 2 * BN * 2 * 8;
  constexpr int smem = smem_main > smem_epi ? smem_main : smem_epi;
  static std::atomic<uint64_t> attr_done{0};
  auto kern = k_conv3x3_halo<WAVES_M, WAVES_N, X2, DROP, MODE, NI>;
  if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), smem, attr_done)) return rc;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int groups = (p.B + g.NB - 1) / g.NB;
  dim3 grid((unsigned)(tiles_n * g.tiles_w * g.tiles_h * groups), p.ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, st, p, g);
  SR3_LAUNCH_CHECK("k_conv3x3_halo");
  return SR3_OK;
}
}  // namespace

// cfg: see halo_cfg_bm / _bn / _split in sr3_common.h.  Returns false when the problem does not fit.
bool halo_geometry(const ConvParams& p, int cfg, HaloGeom* g) {
  if (p.ksize != 3 || p.stride != 1 || cfg < 5 || cfg > 10) return false;
  const int H = p.Ho, W = p.Wo;
  const int BM = halo_cfg_bm(cfg);
  int TW, TH, NB;
  if (W >= 16) { TW = 16; TH = BM / 16; NB = 1; }
  else if (W == 8) { TW = 8; TH = 8; NB = BM / 64; }
  else return false;
  if (BM == 256 && W < 16) return false;
  if (H % TH || W % TW) return false;
  if (NB > 2) return false;
  g->TH = TH; g->TW = TW; g->NB = NB;
  g->log_tw = ilog2(TW); g->log_thw = ilog2(TH * TW);
  g->tiles_w = W / TW; g->tiles_h = H / TH;
  g->HPI = (TH + 2) * (TW + 2);
  g->HP = g->HPI * NB;
  return g->HP <= (BM == 128 ? 200 : 324);
}

int halo_stats_slices(const HaloGeom& g) { return g.NB == 1 ? g.tiles_h * g.tiles_w : 1; }

namespace {
template <int WM, int WN, int MODE, int NI = 2>
int launch_halo_x(const ConvParams& p, const HaloGeom& g, hipStream_t st) {
  if (p.drop_thresh != 0) {       // train-mode block2 convs only
    if constexpr (MODE == 1 || !(WM == 2 && WN == 2)) {
      set_error("conv: this halo tile has no dropout instantiation");
      return SR3_E_UNSUPPORTED;
    } else {
      return p.x2_w ? launch_halo<WM, WN, true, true, 0, NI>(p, g, st) : launch_halo<WM, WN, false, true, 0, NI>(p, g, st);
    }
  }
  return p.x2_w ? launch_halo<WM, WN, true, false, MODE, NI>(p, g, st) : launch_halo<WM, WN, false, false, MODE, NI>(p, g, st);
}
}  // namespace

int conv3x3_halo_forward(const ConvParams& p, int cfg, const HaloGeom& g, hipStream_t st) {
  switch (cfg) {
    case 5: return launch_halo_x<2, 2, 0>(p, g, st);
    case 6:
      // train-mode dropout: the 4-wave 64x64 form of this tile spills ~150 VGPRs with the mask hash in the staging
      // step; the same 256x64 tile on 8 waves of 64x32 does not
      if (p.drop_thresh != 0)
        return p.x2_w ? launch_halo<4, 2, true, true, 0, 1>(p, g, st) : launch_halo<4, 2, false, true, 0, 1>(p, g, st);
      return launch_halo_x<4, 1, 0>(p, g, st);
#ifdef SR3_EXPERIMENTS
    case 7: return launch_halo_x<2, 2, 1>(p, g, st);    // opt-in 3 x bf16 split MFMA (inference only; plan option split_bf16)
    case 8: return launch_halo_x<4, 2, 1, 1>(p, g, st);   // 256x64 on 8 waves of 64x32 (the 4-wave 64x64 form spills)
    case 10: return launch_halo_x<4, 2, 1>(p, g, st);
#else
    case 7: case 8: case 10:      // round 1's split form of the direct kernels: superseded by the Winograd / im2col SPLIT instantiations
      set_error("conv: the split_bf16 halo tiles (7, 8, 10) are an experiment: build with -DSR3_EXPERIMENTS");
      return SR3_E_UNSUPPORTED;
#endif
    case 9: return launch_halo_x<4, 2, 0>(p, g, st);    // 8 waves, one workgroup per CU
  }
  set_error("conv: bad halo tile_cfg %d", cfg);
  return SR3_E_BADARG;
}

}  // namespace sr3
