// 3x3 stride-1 convolution as Winograd F(2x2, 3x3) on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32.
//
// Same op as conv3x3_halo.hip (`Block` = GroupNorm -> Swish -> Conv3x3, model/sr3_modules/unet.py:80-91, Upsample's conv
// :58-65, the skip concat :255) with 2.25x fewer multiplies: on gfx950 the fp32 MFMA runs at the fp32 vector rate
// (157 TF), so the contraction is MFMA-bound and the only way past that roof in fp32 is to multiply less.  Stock
// PyTorch-ROCm does the same for this network (MIOpen picks miopenSp3AsmConv_*_fp32_f2x3 / f3x2 Winograd kernels,
// profiles/r02_torch_rocm_kernel_stats.csv), i.e. this is also the arithmetic the reference itself runs on a GPU.
// fp32 error of F(2x2,3x3) is of the direct convolution's order (transforms with coefficients 0, +-1, +-1/2 only):
// whole-UNet max error vs float64 1.35e-6 against 1.17e-6 for direct fp32 (tests/test_gpu_ops.py).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// i.e. 16 independent GEMMs (one per position (i,j) of the 4x4 transform domain)
//   M_ij[tile][n] = sum_c V_ij[tile][c] * U_ij[n][c],   tile = 2x2 output block, n = output channel, c = input channel.
//
// Workgroup = 8 waves (one per CU: 2 waves / SIMD): 64 Winograd tiles (16x16 output pixels of one image) x 64 output
// channels x all 16 positions (8x8 maps keep the direct kernel).  Wave w owns transform row i = w >> 1 and the two columns
// j = 2 (w & 1), 2 (w & 1) + 1: 2 positions x (64 x 64) outputs = 128 accumulator registers.
// Per 16-channel chunk of the (virtual concat) input:
//   1. the raw input halo ((TH+2) x (TW+2) pixels x 16 channels) is staged ONCE by the whole workgroup, GroupNorm
//      scale/shift + SiLU applied at the store, zero padding after the activation, x2-nearest gather and concat seam
//      resolved in the address (exactly as the halo kernel does);
//   2. every wave builds ITS two V_ij planes from the raw tile (lane = tile; row pass then column pass, 5 float4 adds and
//      6 ds_read_b128 per channel quad) into a wave-private LDS region -- no workgroup barrier between transform and MFMA;
//   3. 64 MFMAs per wave: A fragments from the wave's V planes (ds_read_b128 = 4 k-steps), B fragments = the transformed
//      filters U, which live in global memory in *fragment-major* order (one fully coalesced 1 KB load per wave and
//      fragment, prefetched half a chunk ahead; no LDS staging: a U fragment is only ever used by one wave of the
//      workgroup, so LDS would add nothing but a copy).
// Epilogue: each wave folds its two columns with A (register adds), the rows are combined through LDS in a FIXED order
// (bitwise reproducible), then bias / FiLM / residual / fused GroupNorm statistics / 16-byte NHWC stores as in the halo
// kernel.  Split-K over chunks writes output-domain slabs for k_splitk_reduce.
//
// U = G g G^T is derived from the OHWI weights by k_wino_weights whenever the weights change (plan-level "derived"
// buffer, 16/9 of the 3x3 weights' size); it is never part of a state dict.
#include <stdlib.h>

#include "sr3_common.h"

namespace sr3 {

namespace {
constexpr int WT = 64;          // Winograd tiles per workgroup (16 x 16 output pixels)
constexpr int WBN = 64;         // output channels per workgroup
constexpr int WCK = 16;         // input channels per chunk
constexpr int WRS = 20;         // LDS row stride (floats) of the raw halo and of the V planes: 16 + 4 pad
constexpr int WNT = 512;        // threads (8 waves)
constexpr int WHP = 324;        // raw halo pixels of one tile: 18 x 18
constexpr int WHI = (WHP * 4 + WNT - 1) / WNT;       // raw float4 items per thread (4 channel quads per pixel)
constexpr int WLDT = 36;        // epilogue exchange row stride (32 + 4)
constexpr int W_RAW_F = WHP * WRS;                    // floats per raw buffer (two of them)
constexpr int W_V_F = 8 * 2 * WT * WRS;               // 8 waves x 2 positions x 64 tiles x stride
constexpr int W_EXCH_F = 8 * 2 * 32 * WLDT;           // epilogue: 8 waves x 2 (q) x 32 tiles x stride
constexpr int W_SMEM_MAIN = (2 * W_RAW_F + W_V_F) * 4;
constexpr int W_SMEM_EPI = W_EXCH_F * 4;               // the exchange block doubles as the statistics parking area
constexpr int W_SMEM = W_SMEM_MAIN > W_SMEM_EPI ? W_SMEM_MAIN : W_SMEM_EPI;

// x * sigmoid(x) with the hardware exp2 (v_exp_f32 on x * log2 e): ~1e-7 absolute error on silu, three instructions
// instead of libm expf's twelve -- every staged element pays for this once per output-channel block
__device__ __forceinline__ float silu_w(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// U = G g G^T in fragment-major order:
//   ufrag[cout_blk][chunk][pos = 4 i + j][nblk 2][kk 2][lane 64][4]
//   lane l of fragment (nblk, kk) holds U_pos[n = cout_blk*64 + nblk*32 + (l & 31)][c = chunk*16 + kk*8 + (l >> 5)*4 .. +3]
// (the B operand of v_mfma_f32_32x32x2_f32 for 4 consecutive k-steps), zero outside Cout / Cin.
// One thread per (n, channel quad): 9 float4 loads, 16 float4 stores.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wino_weights(const float* __restrict__ w, int Cout, int Cin, int nchunks,
                                                       int ncb, float* __restrict__ ufrag) {
  const int quads = nchunks * 4;                       // channel quads over the padded Cin
  const long total = (long)ncb * WBN * quads;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(idx % quads);
    const int n = (int)(idx / quads);
    const int c = cq * 4;
    f32x4 g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      g[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (n < Cout && c < Cin) g[t] = *reinterpret_cast<const f32x4*>(w + ((size_t)n * 9 + t) * Cin + c);
    }
    // rows: Gg[i][s], i = 0..3 over r
    f32x4 gg[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const f32x4 g0 = g[0 * 3 + s], g1 = g[1 * 3 + s], g2 = g[2 * 3 + s];
      gg[0][s] = g0;
      gg[1][s] = (g0 + g1 + g2) * 0.5f;
      gg[2][s] = (g0 - g1 + g2) * 0.5f;
      gg[3][s] = g2;
    }
    const int cb = n / WBN, nl = n - cb * WBN;
    const int nblk = nl >> 5;
    const int chunk = c / WCK, cl = c - chunk * WCK;
    const int kk = cl >> 3, hi = (cl >> 2) & 1;
    const int lane = (nl & 31) + 32 * hi;
    float* base = ufrag + ((size_t)(cb * nchunks + chunk) * 16) * 1024 + ((nblk * 2 + kk) * 64 + lane) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 u[4];
      u[0] = gg[i][0];
      u[1] = (gg[i][0] + gg[i][1] + gg[i][2]) * 0.5f;
      u[2] = (gg[i][0] - gg[i][1] + gg[i][2]) * 0.5f;
      u[3] = gg[i][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(base + (size_t)(i * 4 + j) * 1024) = u[j];
    }
  }
}

size_t wino_weight_floats(int Cout, int Cin) {
  const size_t ncb = (Cout + WBN - 1) / WBN, nch = (Cin + WCK - 1) / WCK;
  return ncb * nch * 16 * 1024;
}

int wino_transform_weights(const float* w_ohwi, int Cout, int Cin, float* ufrag, hipStream_t st) {
  if ((Cin & 3) != 0) { set_error("wino: Cin %% 4 != 0"); return SR3_E_UNSUPPORTED; }
  const int ncb = (Cout + WBN - 1) / WBN, nch = (Cin + WCK - 1) / WCK;
  const long total = (long)ncb * WBN * nch * 4;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_wino_weights, dim3(blocks), dim3(256), 0, st, w_ohwi, Cout, Cin, nch, ncb, ufrag);
  SR3_LAUNCH_CHECK("k_wino_weights");
  return SR3_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// DBG (profiling ablations only, env SR3_WINO_DBG; 0 in production): 1 skip the MFMAs, 2 skip the GroupNorm / SiLU arithmetic of
// the staging step, 4 skip the input transforms, 8 skip the epilogue, 16 skip the U loads of the loop, 32 skip the raw staging
// of the loop (global loads + activation + LDS stores)
template <int DBG>
__global__ __launch_bounds__(WNT, 1) void k_conv3x3_wino(const ConvParams p, const WinoGeom g,
                                                         const float* __restrict__ ufrag) {
  extern __shared__ f32x4 smem_v[];
  float* smem = reinterpret_cast<float*>(smem_v);
  float* raw0 = smem;                             // [WHP][WRS] x 2 (double buffered)
  float* raw1 = smem + W_RAW_F;
  float* vbase = smem + 2 * W_RAW_F;              // [8 waves][2][64][WRS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Cin = p.C0 + p.C1;
  const int H = p.Ho, W = p.Wo;
  int bid = blockIdx.x;
  // XCD-aware order: consecutive workgroup ids go round-robin to the 8 XCDs; give each XCD one contiguous range of the
  // (cout block major) tile list so that the U fragments of a cout block stay inside one L2
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const int sp_tiles = g.tiles_w * g.tiles_h * p.B;
  const int cb = bid / sp_tiles;
  int sp = bid - cb * sp_tiles;
  const int tw_i = sp % g.tiles_w;
  sp /= g.tiles_w;
  const int th_i = sp % g.tiles_h;
  const int b0 = sp / g.tiles_h;                  // one image per tile
  const int h0 = th_i * 16, w0 = tw_i * 16;
  constexpr int TWp = 18;

  const int nch = (Cin + WCK - 1) / WCK;
  const int cper = (nch + p.ksplit - 1) / p.ksplit;
  const int c_begin = blockIdx.y * cper;
  const int c_end = min(nch, c_begin + cper);

  // ---- raw staging items of this thread: item j covers halo pixel (tid >> 2) + 128 j, channel quad tid & 3 ----
  const int kq = tid & 3, lrow = tid >> 2;
  int hpix[WHI];
#pragma unroll
  for (int j = 0; j < WHI; ++j) {
    const int hp = lrow + (WNT / 4) * j;
    int pix = -1;
    if (hp < WHP) {
      const int hy = hp / TWp, hx = hp - hy * TWp;
      const int ih = h0 + hy - 1, iw = w0 + hx - 1;
      if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
        pix = (b0 * p.Hs + (ih >> p.ups)) * p.Ws + (iw >> p.ups);
    }
    hpix[j] = pix;
  }
  f32x4 rh[WHI];
  f32x4 ssa, ssb;           // scale/shift of this thread's channel quad
  bool hvalid = false;
  auto load_raw = [&](int chunk) {
    const int c = chunk * WCK + kq * 4;
    hvalid = c < Cin;
    const int ce = hvalid ? c : 0;
    const bool second = ce >= p.C0;
    const float* sp_ = second ? p.src1 : p.src0;
    const int sC = second ? p.C1 : p.C0;
    const int cs = second ? ce - p.C0 : ce;
#pragma unroll
    for (int j = 0; j < WHI; ++j) {
      const int off = hpix[j] >= 0 ? hpix[j] * sC + cs : 0;
      rh[j] = *reinterpret_cast<const f32x4*>(sp_ + off);
    }
    if (p.act != 0) {
      const float* q = p.ss + ((size_t)b0 * Cin + ce) * 2;
      ssa = *reinterpret_cast<const f32x4*>(q);
      ssb = *reinterpret_cast<const f32x4*>(q + 4);
    }
  };
  auto store_raw = [&](float* raw) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < WHI; ++j) {
      const int hp = lrow + (WNT / 4) * j;
      if (hp < WHP) {
        f32x4 v = rh[j];
        if (p.act != 0 && !(DBG & 2)) {
          v.x = fmaf(v.x, ssa.x, ssa.y);
          v.y = fmaf(v.y, ssa.z, ssa.w);
          v.z = fmaf(v.z, ssb.x, ssb.y);
          v.w = fmaf(v.w, ssb.z, ssb.w);
          if (p.act == 2) { v.x = silu_w(v.x); v.y = silu_w(v.y); v.z = silu_w(v.z); v.w = silu_w(v.w); }
        }
        v = (hvalid && hpix[j] >= 0) ? v : zero;
        *reinterpret_cast<f32x4*>(&raw[hp * WRS + kq * 4]) = v;
      }
    }
  };

  // ---- this wave's transform row / columns ----
  const int wi = wave >> 1, wh = wave & 1;
  // rows of the 4x4 patch combined by B^T row wi:  t = d[ra] + sgn * d[rb]
  const int ra = (wi == 0) ? 0 : (wi == 2 ? 2 : 1);
  const int rb = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
  const float rsgn = (wi == 1) ? 1.f : -1.f;
  // lane = tile (8 x 8 tiles of 2x2 outputs): float offset of the two patch rows this wave combines
  const int ty = lane >> 3, tx = lane & 7;
  const int offa = ((2 * ty + ra) * TWp + 2 * tx + wh) * WRS;
  const int offb = ((2 * ty + rb) * TWp + 2 * tx + wh) * WRS;
  const int rot = ty & 1;                           // per-lane channel-quad swap inside a half chunk: spreads the LDS banks
  float* vw = vbase + wave * (2 * WT * WRS);        // this wave's two V planes [2][64][WRS]
  // the transform of one channel quad, split so that its six LDS reads can be in flight across an MFMA block:
  //   t_load issues the reads, t_finish does the row pass (3 FMAs), the column pass (2 adds) and the two LDS writes
  auto t_load = [&](const float* rawbuf, int cq, f32x4 (&da)[3], f32x4 (&db)[3]) {
    if (DBG & 4) return;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      da[s] = *reinterpret_cast<const f32x4*>(rawbuf + offa + s * WRS + cq * 4);
      db[s] = *reinterpret_cast<const f32x4*>(rawbuf + offb + s * WRS + cq * 4);
    }
  };
  auto t_finish = [&](int cq, const f32x4 (&da)[3], const f32x4 (&db)[3]) {
    if (DBG & 4) return;
    f32x4 t[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) t[s] = da[s] + db[s] * rsgn;
    // wh == 0: columns j = 0, 1 from t0, t1, t2:  V0 = t0 - t2, V1 = t1 + t2
    // wh == 1: columns j = 2, 3 from t1, t2, t3:  V2 = t2 - t1, V3 = t1 - t3     (t[] = t1, t2, t3)
    const f32x4 va = wh == 0 ? t[0] - t[2] : t[1] - t[0];
    const f32x4 vb = wh == 0 ? t[1] + t[2] : t[0] - t[2];
    *reinterpret_cast<f32x4*>(vw + (0 * WT + lane) * WRS + cq * 4) = va;
    *reinterpret_cast<f32x4*>(vw + (1 * WT + lane) * WRS + cq * 4) = vb;
  };

  // ---- U fragments: [pj][nblk][kk] float4, straight from global in fragment-major order ----
  f32x4 u[2][2][2];
  const float* ubase = ufrag + (size_t)cb * nch * 16 * 1024 + (size_t)(wi * 4 + wh * 2) * 1024 + lane * 4;
  auto load_u = [&](int chunk, int kk) {             // the four fragments of half a chunk (channels 8 kk .. 8 kk + 7)
    if ((DBG & 16) && chunk != c_begin) return;
    const float* q = ubase + (size_t)chunk * 16 * 1024;
#pragma unroll
    for (int pj = 0; pj < 2; ++pj)
#pragma unroll
      for (int n = 0; n < 2; ++n) u[pj][n][kk] = *reinterpret_cast<const f32x4*>(q + pj * 1024 + (n * 2 + kk) * 256);
  };

  f32x16 acc[2][2][2];          // [pj][mblk][nblk]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;
  const int kh = (lane >> 5) * 4;
  auto mfma_block = [&](int pj, int kk) {            // 16 MFMAs: this wave's position pj, channels 8 kk .. 8 kk + 7
    f32x4 a[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
      a[m] = *reinterpret_cast<const f32x4*>(vw + (pj * WT + m * 32 + (lane & 31)) * WRS + kk * 8 + kh);
    if (DBG & 1) {               // keep the operands live, issue no MFMA
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[pj][m][n][0] += a[m][0] * u[pj][n][kk][0];
      return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[pj][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q], u[pj][n][kk][q], acc[pj][m][n], 0, 0, 0);
  };

  // ---- main loop, software pipelined over half chunks -----------------------------------------------------------
  // Chunk i's raw tile lives in raw[i & 1].  V holds channel quads {0,1} (kk = 0) and {2,3} (kk = 1) of both positions;
  // a half is rebuilt for the NEXT use as soon as its two MFMA blocks are done, with the six LDS reads of every quad
  // issued before an MFMA block and consumed after it:
  //   block A(i): MFMA (pj 0|1, kk 0)   ||  transform half kk 1 of chunk i     (reads raw[i & 1])
  //   barrier                               raw[i & 1] is free -> stage chunk i + 2 into it; chunk i + 1 is visible
  //   block B(i): MFMA (pj 0|1, kk 1)   ||  transform half kk 0 of chunk i + 1 (reads raw[(i + 1) & 1])
  const int nck = c_end - c_begin;
  if (nck > 0) {
    load_raw(c_begin);
    store_raw(raw0);
    if (nck > 1) { load_raw(c_begin + 1); store_raw(raw1); }
    if (nck > 2) load_raw(c_begin + 2);
    load_u(c_begin, 0);
    load_u(c_begin, 1);
    __syncthreads();
    {
      f32x4 da[3], db[3];
      t_load(raw0, rot, da, db);
      t_finish(rot, da, db);
      t_load(raw0, rot ^ 1, da, db);
      t_finish(rot ^ 1, da, db);
    }
    for (int i = 0; i < nck; ++i) {
      float* rcur = (i & 1) ? raw1 : raw0;
      const float* rnext = (i & 1) ? raw0 : raw1;
      const bool more = i + 1 < nck;
      f32x4 da[3], db[3];
      // block A  (sched_barrier: keep the transform's LDS reads ahead of the MFMA block and its arithmetic behind it --
      // left alone the scheduler sinks the reads below the MFMAs and waits on them at once)
      t_load(rcur, 2 + rot, da, db);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      t_finish(2 + rot, da, db);
      t_load(rcur, 2 + (rot ^ 1), da, db);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(1, 0);
      __builtin_amdgcn_sched_barrier(0);
      t_finish(2 + (rot ^ 1), da, db);
      if (more) load_u(c_begin + i + 1, 0);
      __syncthreads();
      if (i + 2 < nck && !(DBG & 32)) {
        store_raw(rcur);
        if (i + 3 < nck) load_raw(c_begin + i + 3);
      }
      // block B
      if (more) t_load(rnext, rot, da, db);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(0, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (more) { t_finish(rot, da, db); t_load(rnext, rot ^ 1, da, db); }
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(1, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (more) { t_finish(rot ^ 1, da, db); load_u(c_begin + i + 1, 1); }
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------------------------
  // fold the two columns of this wave with A (A^T = [1 1 1 0; 0 1 -1 -1]):  P_q = sum_j M_ij A[j][q]
  //   wh == 0 (j = 0, 1): P_0 = M0 + M1, P_1 = M1          wh == 1 (j = 2, 3): P_0 = M2, P_1 = -M2 - M3
  // then Y[p][q] = sum_i A^T[p][i] (P_q(i,0) + P_q(i,1)) through LDS in a FIXED order, one 32-tile x 32-channel block
  // per round (4 rounds).
  if (DBG & 8) {                 // one store per thread keeps the accumulators live
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) s += acc[a][b][c][r];
    p.out[(size_t)bid * WNT + tid] = s;
    return;
  }
  __syncthreads();
  float* exch = smem;                                              // [8 waves][2 q][32][WLDT]
  double* part = reinterpret_cast<double*>(smem);                  // statistics: [512 threads][2 e][8], after the reads
  const bool direct = p.ksplit == 1;
  const bool stats = direct && p.ostat != nullptr;
  const size_t Mtot = (size_t)p.B * H * W;
  float* dst = direct ? p.out : p.partial + (size_t)blockIdx.y * Mtot * p.Cout;
  // final-combine mapping: thread -> (p, q) sub-pixel of the 2x2 block, 16 tile rows x 8 channel quads, two tile halves e
  const int pq = tid >> 7, fp = pq >> 1, fq = pq & 1;
  const int ftile = (tid & 127) >> 3, fcq = tid & 7;
  const int T = g.tiles_h * g.tiles_w;                             // statistics partials per image
  const int tix = th_i * g.tiles_w + tw_i;
  double s1[2][4], s2[2][4];                                       // this thread's channel quad of either 32-channel block
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int k = 0; k < 4; ++k) { s1[a][k] = 0.0; s2[a][k] = 0.0; }
#pragma unroll
  for (int nblk = 0; nblk < 2; ++nblk) {
    const int n = cb * WBN + nblk * 32 + fcq * 4;
    const bool nok = n < p.Cout;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (direct && nok && p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int mblk = 0; mblk < 2; ++mblk) {
      f32x16 p0, p1;
      if (wh == 0) {
        p0 = acc[0][mblk][nblk] + acc[1][mblk][nblk];
        p1 = acc[1][mblk][nblk];
      } else {
        p0 = acc[0][mblk][nblk];
        p1 = -acc[0][mblk][nblk] - acc[1][mblk][nblk];
      }
      // D layout: reg r of lane l -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31
      float* e0 = exch + (wave * 2 + 0) * 32 * WLDT;
      float* e1 = exch + (wave * 2 + 1) * 32 * WLDT;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        e0[row * WLDT + (lane & 31)] = p0[r];
        e1[row * WLDT + (lane & 31)] = p1[r];
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int trow = ftile + 16 * e;                 // tile row inside this 32-tile block
        auto rd = [&](int i, int hh) {
          return *reinterpret_cast<const f32x4*>(exch + (((i * 2 + hh) * 2 + fq) * 32 + trow) * WLDT + fcq * 4);
        };
        const f32x4 r0 = rd(0, 0) + rd(0, 1), r1 = rd(1, 0) + rd(1, 1), r2 = rd(2, 0) + rd(2, 1), r3 = rd(3, 0) + rd(3, 1);
        f32x4 v = fp == 0 ? (r0 + r1) + r2 : (r1 - r2) - r3;
        const int tau = mblk * 32 + trow;                // tile of the 8 x 8 grid
        const int ty = tau >> 3, tx = tau & 7;
        const int b = b0;
        if (b < p.B && nok) {
          const size_t pix = ((size_t)b * H + (h0 + 2 * ty + fp)) * W + (w0 + 2 * tx + fq);
          if (direct) {
            v += bias4;
            if (p.film) v += *reinterpret_cast<const f32x4*>(p.film + (size_t)b * p.film_stride + n);
            if (p.res0) {
              if (n < p.RC0) v += *reinterpret_cast<const f32x4*>(p.res0 + pix * p.RC0 + n);
              else v += *reinterpret_cast<const f32x4*>(p.res1 + pix * p.RC1 + (n - p.RC0));
            }
            if (stats) {
#pragma unroll
              for (int k = 0; k < 4; ++k) { const double dv = (double)v[k]; s1[nblk][k] += dv; s2[nblk][k] += dv * dv; }
            }
          }
          *reinterpret_cast<f32x4*>(dst + pix * p.Cout + n) = v;
        }
      }
      __syncthreads();                                  // every read of the exchange block is complete
    }
  }
  if (stats) {
    // Per-channel sums of this tile's outputs (one image per tile), reduced once, in a fixed order: every thread parks the
    // sums of its two channel quads (one per 32-channel block) in the now free exchange region, then one thread per
    // (channel, sum) walks the 64 threads (4 sub-pixels x 16 tile rows) that share its channel quad.
#pragma unroll
    for (int nb2 = 0; nb2 < 2; ++nb2)
#pragma unroll
      for (int k = 0; k < 4; ++k) { part[(tid * 2 + nb2) * 8 + k] = s1[nb2][k]; part[(tid * 2 + nb2) * 8 + 4 + k] = s2[nb2][k]; }
    __syncthreads();
    if (tid < 128) {
      const int c = tid & 63, which = tid >> 6;              // channel of the 64-wide block, 0 = sum | 1 = sum of squares
      const int nb2 = c >> 5, cq = (c & 31) >> 2, k = c & 3;
      double a = 0.0;
#pragma unroll 8
      for (int src = 0; src < 64; ++src) {
        const int t = (src >> 4) * 128 + (src & 15) * 8 + cq;
        a += part[(t * 2 + nb2) * 8 + which * 4 + k];
      }
      const int nn = cb * WBN + c;
      if (nn < p.Cout) p.ostat[(((size_t)b0 * T + tix) * p.Cout + nn) * 2 + which] = a;
    }
  }
}

// ---- host -----------------------------------------------------------------------------------------------------
namespace {
inline int ilog2x(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
}  // namespace

bool wino_geometry(const ConvParams& p, WinoGeom* g) {
  if (p.ksize != 3 || p.stride != 1) return false;
  const int H = p.Ho, W = p.Wo;
  if (H != (p.Hs << p.ups) || W != (p.Ws << p.ups)) return false;
  if (W < 16 || (W % 16) != 0 || (H % 16) != 0) return false;     // 8x8 maps stay on the direct kernel (no gain measured)
  g->TH = 16; g->TW = 16; g->NB = 1;
  g->twt = 8; g->log_twt = 3;
  g->tpi = 64; g->log_tpi = 6;
  g->tiles_w = W / 16; g->tiles_h = H / 16;
  g->HPI = WHP;
  g->HP = WHP;
  return true;
}
int wino_stats_slices(const WinoGeom& g) { return g.tiles_h * g.tiles_w; }
long wino_workgroups(const ConvParams& p, const WinoGeom& g) {
  return (long)((p.Cout + WBN - 1) / WBN) * g.tiles_w * g.tiles_h * p.B;
}
int wino_chunks(const ConvParams& p) { return (p.C0 + p.C1 + WCK - 1) / WCK; }

int conv3x3_wino_forward(const ConvParams& p, const float* ufrag, hipStream_t st) {
  WinoGeom g;
  if (!wino_geometry(p, &g)) { set_error("conv: the Winograd kernel does not fit this problem"); return SR3_E_UNSUPPORTED; }
  if (!ufrag) { set_error("conv: Winograd needs the transformed weights"); return SR3_E_BADARG; }
  if (p.x2_w || p.drop_thresh != 0) { set_error("conv: the Winograd kernel has no fused 1x1 segment / dropout form"); return SR3_E_UNSUPPORTED; }
  const int nch = wino_chunks(p);
  if (p.ksplit > 1 && (long)(p.ksplit - 1) * ((nch + p.ksplit - 1) / p.ksplit) >= nch) { set_error("conv: ksplit %d leaves an empty split over %d chunks", p.ksplit, nch); return SR3_E_BADARG; }
  dim3 grid((unsigned)wino_workgroups(p, g), p.ksplit);
  static const int dbg = [] { const char* e = getenv("SR3_WINO_DBG"); return e ? atoi(e) : 0; }();
#define SR3_WINO_LAUNCH(D)                                                                                   \
  {                                                                                                          \
    static std::atomic<uint64_t> done{0};                                                                    \
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(k_conv3x3_wino<D>), W_SMEM, done)) return rc;  \
    hipLaunchKernelGGL(k_conv3x3_wino<D>, grid, dim3(WNT), W_SMEM, st, p, g, ufrag);                         \
  }
  switch (dbg) {
    case 0: SR3_WINO_LAUNCH(0) break;
#ifdef SR3_WINO_ABLATIONS
    case 1: SR3_WINO_LAUNCH(1) break;
    case 2: SR3_WINO_LAUNCH(2) break;
    case 4: SR3_WINO_LAUNCH(4) break;
    case 8: SR3_WINO_LAUNCH(8) break;
    case 16: SR3_WINO_LAUNCH(16) break;
    case 32: SR3_WINO_LAUNCH(32) break;
    case 38: SR3_WINO_LAUNCH(38) break;       // MFMA + U + epilogue only
    case 46: SR3_WINO_LAUNCH(46) break;       // MFMA + U only
    case 62: SR3_WINO_LAUNCH(62) break;       // bare MFMA loop
#endif
    default: set_error("conv: SR3_WINO_DBG=%d is not built (compile with -DSR3_WINO_ABLATIONS)", dbg); return SR3_E_BADARG;
  }
#undef SR3_WINO_LAUNCH
  SR3_LAUNCH_CHECK("k_conv3x3_wino");
  return SR3_OK;
}

}  // namespace sr3
