// Weight gradient of every conv of the UNet (autograd of nn.Conv2d in the reference's
// `l_pix.backward()`, model/model.py:54) as a GEMM over pixels on v_mfma_f32_32x32x2_f32:
//     dw[n][tap][c] = sum_{m = (b, oh, ow)} dy[m][n] * a[b, oh*s + r - pad, ow*s + q - pad, c]
// with a = the conv's activated input.  The training driver materialises a = dropout(silu(gn(x0|x1))) once per
// conv (k_apply_act) and passes it as a single source; the generic kernel can also apply the prologue itself and
// folds the nearest x2 upsample / stride into the address (used for the Upsample / Downsample / 1x1 convs).
// Both operands are pixel-major ([m][channels], the natural NHWC order), so a k-step reads its
// fragments with conflict-free ds_read_b32 (lanes along channels).  Workgroup tile TN x TC of one
// filter tap; pixels are walked in chunks of 32, register-prefetched and double-buffered; the pixel
// range is split over gridDim.z and the partial slabs are summed by a deterministic reduce kernel.
#include <algorithm>

#include <string.h>

#include "sr3_common.h"
#include "train.h"

namespace sr3 {

__device__ __forceinline__ float silu_w(float v) { return SR3_SILU(v); }

template <int TN, int TC>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad(const ConvParams p, const float* __restrict__ dy,
                                                        float* __restrict__ slabs, int chunks_per_split, int logW,
                                                        int logHW) {
  constexpr int LDN = TN + 4, LDC = TC + 4;
  constexpr int WN = TN / 2, WC = TC / 2;
  constexpr int MI = WN / 32, NI = WC / 32;
  constexpr int STAGE = 32 * (LDN + LDC);
  extern __shared__ f32x4 smem_v[];
  float* smem = reinterpret_cast<float*>(smem_v);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Cin = p.C0 + p.C1;
  const int taps = p.ksize * p.ksize;
  const int pad = p.ksize / 2;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.B * HoWo;
  const int Hi = p.Hs << p.ups, Wi = p.Ws << p.ups;
  const int tiles_c = (Cin + TC - 1) / TC;
  const int tile_n = blockIdx.x / tiles_c, tile_c = blockIdx.x - tile_n * tiles_c;
  const int tap = blockIdx.y;
  const int fr = tap / p.ksize, fs = tap - fr * p.ksize;
  const int nchunks = (M + 31) / 32;
  const int ch0 = blockIdx.z * chunks_per_split;
  const int ch1 = min(nchunks, ch0 + chunks_per_split);

  const int lq = tid & 31, lpx = tid >> 5;        // loader: channel quad, pixel row (0..7) + 8 i
  f32x4 ry[4], ra[4], sa[4], sb[4];
  bool yok[4], aok[4];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  auto load = [&](int chunk) {
    const int n = tile_n * TN + lq * 4;
    const int c = tile_c * TC + lq * 4;
    const bool nv = (lq * 4 < TN) && n < p.Cout;
    const bool cv = (lq * 4 < TC) && c < Cin;
    const int ce = cv ? c : 0;
    const bool second = ce >= p.C0;
    const float* sp = second ? p.src1 : p.src0;
    const int sC = second ? p.C1 : p.C0;
    const int cs = second ? ce - p.C0 : ce;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = chunk * 32 + lpx + 8 * i;
      const bool mv = m < M;
      const int me = mv ? m : 0;
      int b, oh, ow;
      if (logW >= 0) {            // power-of-two feature maps (every real config): shifts instead of divisions
        b = me >> logHW;
        const int rem = me & (HoWo - 1);
        oh = rem >> logW; ow = rem & (p.Wo - 1);
      } else {
        b = me / HoWo;
        const int rem = me - b * HoWo;
        oh = rem / p.Wo; ow = rem - oh * p.Wo;
      }
      yok[i] = mv && nv;
      ry[i] = *reinterpret_cast<const f32x4*>(dy + (yok[i] ? me * p.Cout + n : 0));
      const int ih = oh * p.stride + fr - pad, iw = ow * p.stride + fs - pad;
      const bool ok = mv && cv && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi;
      aok[i] = ok;
      const int pix = (b * p.Hs + (ih >> p.ups)) * p.Ws + (iw >> p.ups);
      ra[i] = *reinterpret_cast<const f32x4*>(sp + (ok ? pix * sC + cs : 0));
      if (p.act != 0) {
        const float* q = p.ss + (b * Cin + ce) * 2;
        sa[i] = *reinterpret_cast<const f32x4*>(q);
        sb[i] = *reinterpret_cast<const f32x4*>(q + 4);
      }
    }
  };
  auto store = [&](int st) {
    float* Ys = smem + st * STAGE;
    float* As = Ys + 32 * LDN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = lpx + 8 * i;
      if (lq * 4 < TN) *reinterpret_cast<f32x4*>(&Ys[px * LDN + lq * 4]) = yok[i] ? ry[i] : zero;
      if (lq * 4 < TC) {
        f32x4 v = ra[i];
        if (p.act != 0) {
          v.x = fmaf(v.x, sa[i].x, sa[i].y);
          v.y = fmaf(v.y, sa[i].z, sa[i].w);
          v.z = fmaf(v.z, sb[i].x, sb[i].y);
          v.w = fmaf(v.w, sb[i].z, sb[i].w);
          if (p.act == 2) { v.x = silu_w(v.x); v.y = silu_w(v.y); v.z = silu_w(v.z); v.w = silu_w(v.w); }
        }
        *reinterpret_cast<f32x4*>(&As[px * LDC + lq * 4]) = aok[i] ? v : zero;
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

  const int wave_n = wave >> 1, wave_c = wave & 1;
  const int ncol = wave_n * WN + (lane & 31);
  const int ccol = wave_c * WC + (lane & 31);
  const int khalf = lane >> 5;

  if (ch0 < ch1) {
    load(ch0);
    store(0);
    __syncthreads();
    for (int ch = ch0; ch < ch1; ++ch) {
      const int cur = (ch - ch0) & 1;
      const bool more = ch + 1 < ch1;
      if (more) load(ch + 1);
      const float* Ys = smem + cur * STAGE;
      const float* As = Ys + 32 * LDN;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const int row = 2 * kk + khalf;
        float a[MI], bq[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = Ys[row * LDN + ncol + 32 * i];
#pragma unroll
        for (int j = 0; j < NI; ++j) bq[j] = As[row * LDC + ccol + 32 * j];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[j], acc[i][j], 0, 0, 0);
      }
      if (more) store(cur ^ 1);
      __syncthreads();
    }
  }

  float* dst = slabs + (size_t)blockIdx.z * p.Cout * taps * Cin;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int c = tile_c * TC + wave_c * WC + 32 * j + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = tile_n * TN + wave_n * WN + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < p.Cout && c < Cin) dst[((size_t)n * taps + tap) * Cin + c] = acc[i][j][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same GEMM on v_mfma_f32_32x32x16_bf16 with 3-way split fp32 operands (round 5; plan option wgrad_split, default ON, used for
// the layers with more than 64 channels on both sides): six bf16 products per fp32 product, fp32 accumulation -- the arithmetic of the split conv kernels.
// The contraction runs over PIXELS, which are the rows of both NHWC operands, while a bf16 MFMA operand wants 8 consecutive k per
// lane: the staging step transposes.  A unit of staging = 8 consecutive pixels (one k-half of a 16-pixel k-step) x 4 channels:
// eight 16-byte loads, per channel one split3x8 of the eight pixel values, three 16-byte LDS writes into
//   plane[3][k-step 2][k-half 2][row][8 bf16]        (row = output channel n for dy, input channel c for the activations)
// so that a lane's MFMA operand is ONE conflict-free ds_read_b128 (lanes along rows).  One tap per workgroup (blockIdx.y), so
// the tap shift lives in the loader's pixel address and nothing has to be re-aligned; 3x3 stride-1 layers run here too under
// wgrad_split (the 9-tap kernel's shared halo would need funnel-shifted fragments).  Single LDS stage,
// the next chunk's loads in flight in registers across the MFMAs.  TN = TC = 128: one unit per thread and operand pair;
// 64 x 64: units of 4 pixels (two threads fill one fragment with 8-byte writes).
// ---------------------------------------------------------------------------------------------------
template <int TN, int TC, bool ACT>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad_split(const ConvParams p, const float* __restrict__ dy,
                                                              float* __restrict__ slabs, int chunks_per_split, int logW,
                                                              int logHW) {
  constexpr int WN = TN / 2, WC = TC / 2;
  constexpr int MI = WN / 32, NI = WC / 32;
  constexpr int PX = (TN == 128) ? 8 : 4;                 // pixels per staging unit
  constexpr int ROWS = TN + TC;                           // fragment rows per (k-step, k-half): dy rows then activation rows
  extern __shared__ f32x4 smem_v[];
  __bf16* planes = reinterpret_cast<__bf16*>(smem_v);     // [plane 3][ks 2][kh 2][ROWS][8]
  constexpr int PLANE = 4 * ROWS * 8;                     // bf16 elements per plane

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Cin = p.C0 + p.C1;
  const int taps = p.ksize * p.ksize;
  const int pad = p.ksize / 2;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.B * HoWo;
  const int Hi = p.Hs << p.ups, Wi = p.Ws << p.ups;
  const int tiles_c = (Cin + TC - 1) / TC;
  const int tile_n = blockIdx.x / tiles_c, tile_c = blockIdx.x - tile_n * tiles_c;
  const int tap = blockIdx.y;
  const int fr = tap / p.ksize, fs = tap - fr * p.ksize;
  const int nchunks = (M + 31) / 32;
  const int ch0 = blockIdx.z * chunks_per_split;
  const int ch1 = min(nchunks, ch0 + chunks_per_split);

  // staging unit of this thread: threads 0-127 (waves 0, 1) stage dy, threads 128-255 the activations; within an operand the unit
  // is (channel quad uq, pixel group ug): TN = 128: 32 quads x 4 groups of 8 pixels, TN = 64: 16 quads x 8 groups of 4 pixels
  const bool is_act = tid >= 128;
  const int u = tid & 127;
  constexpr int QUADS = TN / 4;                           // (TN == TC)
  const int uq = u % QUADS, ug = u / QUADS;               // quad, pixel group (8 px: 0..3; 4 px: 0..7)
  f32x4 rv[PX], sa[ACT ? PX : 1], sb[ACT ? PX : 1];       // (ACT: the GroupNorm pairs of the unit's pixels travel with the prefetch)
  bool rok[PX];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  auto load = [&](int chunk) {
    const int n = tile_n * TN + uq * 4;
    const int c = tile_c * TC + uq * 4;
    const bool nv = n < p.Cout, cv = c < Cin;
    const int ce = cv ? c : 0;
    const bool second = ce >= p.C0;
    const float* sp = second ? p.src1 : p.src0;
    const int sC = second ? p.C1 : p.C0;
    const int cs = second ? ce - p.C0 : ce;
    // power-of-two maps at least PX wide: the unit's PX pixels lie in one image row -- decode the first, step the column (round 6: the
    // per-pixel decode was a third of this kernel's 6 VALU instructions per MFMA)
    int ub = 0, uoh = 0, uow = 0;
    const bool rowunit = logW >= 3 && PX == 8;
    if (rowunit) {
      const int m0 = min(chunk * 32 + ug * PX, M - PX);
      ub = m0 >> logHW;
      const int rem = m0 & (HoWo - 1);
      uoh = rem >> logW; uow = rem & (p.Wo - 1);
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      const int m = chunk * 32 + ug * PX + i;
      const bool mv = m < M;
      const int me = mv ? m : 0;
      int b, oh, ow;
      if (rowunit) {
        b = ub; oh = uoh; ow = uow + i;
      } else if (logW >= 0) {
        b = me >> logHW;
        const int rem = me & (HoWo - 1);
        oh = rem >> logW; ow = rem & (p.Wo - 1);
      } else {
        b = me / HoWo;
        const int rem = me - b * HoWo;
        oh = rem / p.Wo; ow = rem - oh * p.Wo;
      }
      if (!is_act) {
        rok[i] = mv && nv;
        rv[i] = *reinterpret_cast<const f32x4*>(dy + (rok[i] ? me * p.Cout + n : 0));
      } else {
        const int ih = oh * p.stride + fr - pad, iw = ow * p.stride + fs - pad;
        const bool ok = mv && cv && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi;
        rok[i] = ok;
        const int pix = (b * p.Hs + (ih >> p.ups)) * p.Ws + (iw >> p.ups);
        rv[i] = *reinterpret_cast<const f32x4*>(sp + (ok ? pix * sC + cs : 0));
        if constexpr (ACT) {
          const float* q = p.ss + (b * Cin + ce) * 2;
          sa[i] = *reinterpret_cast<const f32x4*>(q);
          sb[i] = *reinterpret_cast<const f32x4*>(q + 4);
        }
      }
    }
  };
  auto store = [&]() {
    f32x4 v[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      v[i] = rv[i];
      if constexpr (ACT) if (is_act) {
        v[i].x = fmaf(v[i].x, sa[i].x, sa[i].y);
        v[i].y = fmaf(v[i].y, sa[i].z, sa[i].w);
        v[i].z = fmaf(v[i].z, sb[i].x, sb[i].y);
        v[i].w = fmaf(v[i].w, sb[i].z, sb[i].w);
        if (p.act == 2) { v[i].x = silu_w(v[i].x); v[i].y = silu_w(v[i].y); v[i].z = silu_w(v[i].z); v[i].w = silu_w(v[i].w); }
      }
      if (!rok[i]) v[i] = zero;
    }
    // k position of the unit's pixels inside the chunk: pixel index ug * PX + i = 16 ks + 8 kh + e
    const int p0 = ug * PX;
    const int ks = p0 >> 4, kh = (p0 >> 3) & 1, e0 = p0 & 7;          // (PX = 8: e0 = 0; PX = 4: e0 = 0 or 4)
    // fragment row of channel 4 uq + chn = chn * QUADS + uq (round 6): the lanes of a write -- consecutive uq -- hit consecutive 16-byte rows
    // (row 4 uq + chn put them 64 bytes apart: four-way bank conflicts, 60 % of the kernel's LDS cycles); the epilogue un-permutes
    const int row0 = (is_act ? TN : 0) + uq;
#pragma unroll
    for (int chn = 0; chn < 4; ++chn) {
      __bf16* dstp = planes + ((size_t)(ks * 2 + kh) * ROWS + row0 + chn * QUADS) * 8 + e0;
      if constexpr (PX == 8) {
        const f32x4 lo = {v[0][chn], v[1][chn], v[2][chn], v[3][chn]}, hi = {v[4][chn], v[5][chn], v[6][chn], v[7][chn]};
        bf16x8 h, m, l;
        split3x8(lo, hi, h, m, l);
        *reinterpret_cast<bf16x8*>(dstp) = h;
        *reinterpret_cast<bf16x8*>(dstp + PLANE) = m;
        *reinterpret_cast<bf16x8*>(dstp + 2 * PLANE) = l;
      } else {
        const f32x4 q4 = {v[0][chn], v[1][chn], v[2][chn], v[3][chn]};
        bf16x4 h, m, l;
        split3(q4, h, m, l);
        *reinterpret_cast<bf16x4*>(dstp) = h;
        *reinterpret_cast<bf16x4*>(dstp + PLANE) = m;
        *reinterpret_cast<bf16x4*>(dstp + 2 * PLANE) = l;
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

  const int wave_n = wave >> 1, wave_c = wave & 1;
  const int nrow = wave_n * WN + (lane & 31);             // fragment row of this lane's dy operand (+ 32 i)
  const int crow = TN + wave_c * WC + (lane & 31);        // ... of its activation operand (+ 32 j)
  const int khl = lane >> 5;

  if (ch0 < ch1) {
    load(ch0);
    store();
    __syncthreads();
    for (int ch = ch0; ch < ch1; ++ch) {
      const bool more = ch + 1 < ch1;
      if (more) load(ch + 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a[MI][3], bq[NI][3];
        const __bf16* base = planes + (size_t)(ks * 2 + khl) * ROWS * 8;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
          for (int i = 0; i < MI; ++i) a[i][pl] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE + (nrow + 32 * i) * 8);
#pragma unroll
          for (int j = 0; j < NI; ++j) bq[j][pl] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE + (crow + 32 * j) * 8);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) mfma_split6(a[i], bq[j], acc[i][j]);
      }
      __syncthreads();                       // every fragment of this chunk is read
      if (more) store();
      __syncthreads();
    }
  }

  float* dst = slabs + (size_t)blockIdx.z * p.Cout * taps * Cin;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int cr = wave_c * WC + 32 * j + (lane & 31);                       // fragment row -> channel: 4 (row % QUADS) + row / QUADS
      const int c = tile_c * TC + 4 * (cr % QUADS) + cr / QUADS;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nr = wave_n * WN + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int n = tile_n * TN + 4 * (nr % QUADS) + nr / QUADS;
        if (n < p.Cout && c < Cin) dst[((size_t)n * taps + tap) * Cin + c] = acc[i][j][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------------
// 9-tap variant for the 3x3 stride-1 convs (the bulk of the weight-gradient FLOPs).  A workgroup owns a
// 64 x 64 (n, c) tile of ALL nine taps: per 32-pixel chunk (a row segment, or 2-4 whole rows of a narrow
// map) it stages dOut[32][64] and the (rows+2) x (cols+2) halo of the materialised activated input ONCE and
// every tap reads its B fragments from the halo at a shifted pixel -- 144 MFMAs per wave per barrier and
// 3.5x fewer staged bytes per MFMA than one tap at a time.  a: [B,H,W,Cin] single source (k_apply_act).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_conv_wgrad9(const float* __restrict__ a, int Cin,
                                                         const float* __restrict__ dy, int Cout, int B, int H, int W,
                                                         int logW, float* __restrict__ slabs, int chunks_per_split) {
  constexpr int T = 64, LD = 68;
  constexpr int HPX_MAX = 102;                       // 3 x 34 halo pixels (W >= 32); 4 x 18, 6 x 10 for W = 16, 8
  constexpr int HI = (HPX_MAX * 16 + 255) / 256;     // halo float4 items per thread (7)
  constexpr int STAGE = (32 + HPX_MAX) * LD;
  extern __shared__ f32x4 smem_v[];
  float* smem = reinterpret_cast<float*>(smem_v);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Wc = W < 32 ? W : 32;                    // chunk = RPC rows x Wc columns = 32 pixels
  const int logWc = logW < 5 ? logW : 5;
  const int RPC = 32 >> logWc;
  const int HWp = Wc + 2, HR = RPC + 2, HPX = HR * HWp;
  const int cpr = W >> logWc;                        // chunks per image row (W >= 32) else 1
  const int chunks_per_img = (H * W) >> 5;
  const int nchunks = B * chunks_per_img;
  const int tiles_c = (Cin + T - 1) / T;
  const int tile_n = blockIdx.x / tiles_c, tile_c = blockIdx.x - tile_n * tiles_c;
  const int ch0 = blockIdx.z * chunks_per_split;
  const int ch1 = min(nchunks, ch0 + chunks_per_split);

  // loader items (fixed per thread): halo pixel (tid>>4) + 16 j, channel quad tid & 15
  const int lq = tid & 15, lp = tid >> 4;
  int hy[HI], hx[HI];
#pragma unroll
  for (int j = 0; j < HI; ++j) {
    const int hp = lp + 16 * j;
    hy[j] = hp < HPX ? hp / HWp : -1000;
    hx[j] = hp < HPX ? hp - (hp / HWp) * HWp : 0;
  }
  f32x4 rh[HI], ry[2];
  bool hok[HI], yok[2];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  auto load = [&](int chunk) {
    const int b = chunk / chunks_per_img;
    const int ci = chunk - b * chunks_per_img;
    const int oh0 = (W >= 32) ? ci / cpr : ci * RPC;
    const int ow0 = (W >= 32) ? (ci - (ci / cpr) * cpr) * 32 : 0;
    const int c = tile_c * T + lq * 4;
    const int n = tile_n * T + lq * 4;
    const bool cv = c < Cin, nv = n < Cout;
#pragma unroll
    for (int j = 0; j < HI; ++j) {
      const int ih = oh0 + hy[j] - 1, iw = ow0 + hx[j] - 1;
      const bool ok = cv && hy[j] >= 0 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      hok[j] = ok;
      rh[j] = *reinterpret_cast<const f32x4*>(a + (ok ? ((b * H + ih) * W + iw) * Cin + c : 0));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int px = lp + 16 * i;                    // 0..31
      const int oh = oh0 + (px >> logWc), ow = ow0 + (px & (Wc - 1));
      yok[i] = nv;
      ry[i] = *reinterpret_cast<const f32x4*>(dy + (nv ? ((b * H + oh) * W + ow) * Cout + n : 0));
    }
  };
  auto store = [&](int st) {
    float* Ys = smem + st * STAGE;
    float* Hs = Ys + 32 * LD;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&Ys[(lp + 16 * i) * LD + lq * 4]) = yok[i] ? ry[i] : zero;
#pragma unroll
    for (int j = 0; j < HI; ++j) {
      const int hp = lp + 16 * j;
      if (hp < HPX) *reinterpret_cast<f32x4*>(&Hs[hp * LD + lq * 4]) = hok[j] ? rh[j] : zero;
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int wave_n = wave >> 1, wave_c = wave & 1;
  const int ncol = wave_n * 32 + (lane & 31);
  const int ccol = wave_c * 32 + (lane & 31);
  const int khalf = lane >> 5;

  if (ch0 < ch1) {
    load(ch0);
    store(0);
    __syncthreads();
    for (int ch = ch0; ch < ch1; ++ch) {
      const int cur = (ch - ch0) & 1;
      const bool more = ch + 1 < ch1;
      if (more) load(ch + 1);
      const float* Ys = smem + cur * STAGE;
      const float* Hs = Ys + 32 * LD;
#pragma unroll 4
      for (int kk = 0; kk < 16; ++kk) {
        const int px = 2 * kk + khalf;
        const float av = Ys[px * LD + ncol];
        const int hbase = (px >> logWc) * HWp + (px & (Wc - 1));      // halo pixel of tap (0,0)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float bv = Hs[(hbase + (t / 3) * HWp + (t % 3)) * LD + ccol];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
      if (more) store(cur ^ 1);
      __syncthreads();
    }
  }
  float* dst = slabs + (size_t)blockIdx.z * Cout * 9 * Cin;
  const int c = tile_c * T + wave_c * 32 + (lane & 31);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = tile_n * T + wave_n * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (n < Cout && c < Cin) dst[((size_t)n * 9 + t) * Cin + c] = acc[t][r];
    }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ slabs, int msplit, size_t n4,
                                                       float* __restrict__ dw) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    // double accumulation: a batch can hold samples whose gradient terms are orders of magnitude larger than the
    // others' (noise levels near 0); summing their slabs with the rest in fp32 costs the small ones their low bits
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    auto slab = [&](int s) { return *reinterpret_cast<const f32x4*>(slabs + ((size_t)s * n4 + i) * 4); };
    int s = 0;
    for (; s + 4 <= msplit; s += 4) {      // four slabs in flight (same order of additions; one load per iteration waited for each in turn)
      const f32x4 v0 = slab(s), v1 = slab(s + 1), v2 = slab(s + 2), v3 = slab(s + 3);
      a0 += (double)v0.x; a1 += (double)v0.y; a2 += (double)v0.z; a3 += (double)v0.w;
      a0 += (double)v1.x; a1 += (double)v1.y; a2 += (double)v1.z; a3 += (double)v1.w;
      a0 += (double)v2.x; a1 += (double)v2.y; a2 += (double)v2.z; a3 += (double)v2.w;
      a0 += (double)v3.x; a1 += (double)v3.y; a2 += (double)v3.z; a3 += (double)v3.w;
    }
    for (; s < msplit; ++s) {
      const f32x4 v = slab(s);
      a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
    }
    const f32x4 r = {(float)a0, (float)a1, (float)a2, (float)a3};
    *reinterpret_cast<f32x4*>(dw + i * 4) = r;
  }
}

namespace {
inline int cdivw(int a, int b) { return (a + b - 1) / b; }
inline int ilog2w(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }
// the 9-tap kernel covers 3x3 stride-1 non-upsampled single-source convs on power-of-two maps with W >= 8
bool wgrad9_ok(const ConvParams& c) {
  if (c.ksize != 3 || c.stride != 1 || c.ups != 0 || c.C1 != 0 || c.act != 0) return false;
  const int lw = ilog2w(c.Wo);
  if (lw < 3 || c.Ho != c.Hs || c.Wo != c.Ws) return false;
  const int Wc = c.Wo < 32 ? c.Wo : 32;
  const int rpc = 32 / Wc;
  return c.Ho % rpc == 0 && (c.Ho * c.Wo) % 32 == 0;
}
// plan option wgrad_split: the one-tap-per-workgroup split kernel on its 128 x 128 tile, i.e. for every layer with more than 64
// input and output channels (3x3 layers included); the 64-channel layers keep their kernels (the 64 x 64 split tile stages as
// many operand values per MFMA as it saves: 292 VALU instructions per 12 MFMAs)
bool wgrad_use_split(const ConvParams& c) { return c.wgrad_split && c.Cout > 64 && c.C0 + c.C1 > 64; }
void wgrad_geometry(const ConvParams& c, int* tn, int* tc, int* msplit, int* cps) {
  const int Cin = c.C0 + c.C1;
  const int taps = c.ksize * c.ksize;
  if (wgrad9_ok(c) && !wgrad_use_split(c)) {
    *tn = 64; *tc = 64;
    const long tiles = (long)cdivw(c.Cout, 64) * cdivw(Cin, 64);
    const int nchunks = c.B * c.Ho * c.Wo / 32;
    long ms = (512 + tiles - 1) / tiles;
    const long cap = nchunks / 2 > 1 ? nchunks / 2 : 1;
    if (ms > cap) ms = cap;
    if (ms < 1) ms = 1;
    int per = cdivw(nchunks, (int)ms);
    *msplit = cdivw(nchunks, per);
    *cps = per;
    return;
  }
  const bool small = c.Cout <= 64 || Cin <= 64;
  *tn = small ? 64 : 128;
  *tc = small ? 64 : 128;
  const long tiles = (long)cdivw(c.Cout, *tn) * cdivw(Cin, *tc) * taps;
  const int nchunks = cdivw(c.B * c.Ho * c.Wo, 32);
  long ms = (1024 + tiles - 1) / tiles;
  const long cap = nchunks / 4 > 1 ? nchunks / 4 : 1;
  if (ms > cap) ms = cap;
  if (ms < 1) ms = 1;
  int per = cdivw(nchunks, (int)ms);
  ms = cdivw(nchunks, per);
  *msplit = (int)ms;
  *cps = per;
}

template <int TN, int TC>
int launch_wgrad(const WgradParams& p, int msplit, int cps, hipStream_t st) {
  constexpr int smem = 2 * 32 * (TN + 4 + TC + 4) * 4;
  static std::atomic<uint64_t> attr_done{0};
  auto kern = k_conv_wgrad<TN, TC>;
  if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), smem, attr_done)) return rc;
  const int Cin = p.c.C0 + p.c.C1;
  const int taps = p.c.ksize * p.c.ksize;
  dim3 grid(cdivw(p.c.Cout, TN) * cdivw(Cin, TC), taps, msplit);
  auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
  int logW = lg(p.c.Wo), logHW = lg(p.c.Ho * p.c.Wo);
  if (logW < 0 || logHW < 0) logW = logHW = -1;
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p.c, p.dy, msplit > 1 ? p.slabs : p.dw, cps, logW, logHW);
  SR3_LAUNCH_CHECK("k_conv_wgrad");
  return SR3_OK;
}
template <int TN, int TC>
int launch_wgrad_split(const WgradParams& p, int msplit, int cps, hipStream_t st) {
  constexpr int smem = 3 * 4 * (TN + TC) * 8 * 2;         // three planes of [ks 2][kh 2][TN + TC rows][8 bf16]
  static std::atomic<uint64_t> attr_done{0};
  auto kern = p.c.act != 0 ? k_conv_wgrad_split<TN, TC, true> : k_conv_wgrad_split<TN, TC, false>;
  if (int rc = ensure_max_lds(reinterpret_cast<const void*>(k_conv_wgrad_split<TN, TC, true>), smem, attr_done)) return rc;
  static std::atomic<uint64_t> attr_done2{0};
  if (int rc = ensure_max_lds(reinterpret_cast<const void*>(k_conv_wgrad_split<TN, TC, false>), smem, attr_done2)) return rc;
  const int Cin = p.c.C0 + p.c.C1;
  const int taps = p.c.ksize * p.c.ksize;
  dim3 grid(cdivw(p.c.Cout, TN) * cdivw(Cin, TC), taps, msplit);
  auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
  int logW = lg(p.c.Wo), logHW = lg(p.c.Ho * p.c.Wo);
  if (logW < 0 || logHW < 0) logW = logHW = -1;
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p.c, p.dy, msplit > 1 ? p.slabs : p.dw, cps, logW, logHW);
  SR3_LAUNCH_CHECK("k_conv_wgrad_split");
  return SR3_OK;
}
}  // namespace

size_t wgrad_slab_bytes(const ConvParams& c, int* msplit_out) {
  // the slab size covers BOTH geometries (the plan sizes its scratch before it knows the option), msplit is the one the launch uses
  size_t bytes = 0;
  ConvParams v = c;
  for (int mode = 0; mode < 2; ++mode) {
    v.wgrad_split = mode;
    int tn, tc, ms, cps;
    wgrad_geometry(v, &tn, &tc, &ms, &cps);
    if (mode == (c.wgrad_split ? 1 : 0) && msplit_out) *msplit_out = ms;
    if (ms > 1) bytes = std::max(bytes, (size_t)ms * c.Cout * c.ksize * c.ksize * (c.C0 + c.C1) * sizeof(float));
  }
  return bytes;
}

int conv_wgrad(const WgradParams& p, hipStream_t st) {
  const ConvParams& c = p.c;
  const int Cin = c.C0 + c.C1;
  if ((c.C0 & 3) || (c.C1 & 3) || (c.Cout & 3)) { set_error("wgrad: channel counts must be multiples of 4"); return SR3_E_UNSUPPORTED; }
  if ((double)c.B * c.Ho * c.Wo * c.Cout >= 2147483647.0 || (double)c.B * c.Hs * c.Ws * (c.C0 > c.C1 ? c.C0 : c.C1) >= 2147483647.0) {
    set_error("wgrad: tensor exceeds 2^31 elements");
    return SR3_E_UNSUPPORTED;
  }
  if (c.act != 0 && !c.ss) { set_error("wgrad: act needs ss"); return SR3_E_BADARG; }
  int tn, tc, ms, cps;
  wgrad_geometry(c, &tn, &tc, &ms, &cps);
  if (ms != p.msplit) { set_error("wgrad: msplit mismatch (%d vs %d)", ms, p.msplit); return SR3_E_BADARG; }
  if (ms > 1 && !p.slabs) { set_error("wgrad: slabs required"); return SR3_E_BADARG; }
  int rc;
  if (wgrad_use_split(c)) {
    rc = launch_wgrad_split<128, 128>(p, ms, cps, st);
  } else if (wgrad9_ok(c)) {
    constexpr int smem9 = 2 * (32 + 102) * 68 * 4;
    static std::atomic<uint64_t> attr9_done{0};
    if (int rc9 = ensure_max_lds(reinterpret_cast<const void*>(k_conv_wgrad9), smem9, attr9_done)) return rc9;
    dim3 grid(cdivw(c.Cout, 64) * cdivw(Cin, 64), 1, ms);
    hipLaunchKernelGGL(k_conv_wgrad9, grid, dim3(256), smem9, st, c.src0, Cin, p.dy, c.Cout, c.B, c.Ho, c.Wo, ilog2w(c.Wo),
                       ms > 1 ? p.slabs : p.dw, cps);
    SR3_LAUNCH_CHECK("k_conv_wgrad9");
    rc = SR3_OK;
  } else {
    rc = tn == 128 ? launch_wgrad<128, 128>(p, ms, cps, st) : launch_wgrad<64, 64>(p, ms, cps, st);
  }
  if (rc) return rc;
  if (ms > 1) {
    const size_t n4 = (size_t)c.Cout * c.ksize * c.ksize * Cin / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)blocks), dim3(256), 0, st, p.slabs, ms, n4, p.dw);
    SR3_LAUNCH_CHECK("k_wgrad_reduce");
  }
  return SR3_OK;
}

}  // namespace sr3

// ---- per-op entry (include/sr3_mi355x.h): the weight gradient of one convolution, for op-level tests of every kernel of this file ----
namespace {
void wgrad_fill(sr3::ConvParams& c, const float* src0, int C0, const float* src1, int C1, int B, int Hs, int Ws, int ups, int stride, int ksize,
                int Cout, const float* ss, int act, int split) {
  memset(&c, 0, sizeof(c));
  c.src0 = src0; c.src1 = src1; c.C0 = C0; c.C1 = src1 ? C1 : 0; c.B = B; c.Hs = Hs; c.Ws = Ws; c.ups = ups; c.stride = stride; c.ksize = ksize;
  const int pad = ksize / 2;
  c.Ho = ((Hs << ups) + 2 * pad - ksize) / stride + 1;
  c.Wo = ((Ws << ups) + 2 * pad - ksize) / stride + 1;
  c.Cout = Cout; c.ss = ss; c.act = act; c.ksplit = 1; c.wgrad_split = split;
}
}  // namespace
extern "C" size_t sr3_conv_wgrad_scratch_bytes(int B, int Hs, int Ws, int ups, int stride, int ksize, int C0, int C1, int Cout, int split) {
  sr3::ConvParams c;
  wgrad_fill(c, nullptr, C0, C1 ? reinterpret_cast<const float*>(1) : nullptr, C1, B, Hs, Ws, ups, stride, ksize, Cout, nullptr, 0, split);
  return sr3::wgrad_slab_bytes(c, nullptr);
}
extern "C" int sr3_conv_wgrad_f32(const float* src0, int C0, const float* src1, int C1, int B, int Hs, int Ws, int ups, int stride, int ksize,
                                  int Cout, const float* ss, int act, const float* dy, float* dw_ohwi, int split, void* scratch,
                                  size_t scratch_bytes, void* stream) {
  if (!src0 || !dy || !dw_ohwi) { sr3::set_error("null argument"); return SR3_E_BADARG; }
  if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || (ups != 0 && ups != 1)) { sr3::set_error("wgrad: ksize 1 | 3, stride 1 | 2, ups 0 | 1"); return SR3_E_UNSUPPORTED; }
  sr3::WgradParams wp;
  wgrad_fill(wp.c, src0, C0, src1, C1, B, Hs, Ws, ups, stride, ksize, Cout, ss, act, split);
  wp.dy = dy; wp.dw = dw_ohwi; wp.slabs = static_cast<float*>(scratch);
  const size_t need = sr3::wgrad_slab_bytes(wp.c, &wp.msplit);
  if (wp.msplit > 1 && (!scratch || scratch_bytes < need)) { sr3::set_error("wgrad: scratch too small (%zu < %zu)", scratch_bytes, need); return SR3_E_NOMEM; }
  return sr3::conv_wgrad(wp, static_cast<hipStream_t>(stream));
}
