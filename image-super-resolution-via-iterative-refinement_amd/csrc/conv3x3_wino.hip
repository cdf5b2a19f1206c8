// 3x3 stride-1 convolution as Winograd F(2x2, 3x3) on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32.
//
// Same op as conv3x3_halo.hip (`Block` = GroupNorm -> Swish -> Conv3x3, model/sr3_modules/unet.py:80-91, Upsample's conv
// :58-65, the skip concat :255) with 2.25x fewer multiplies: on gfx950 the fp32 MFMA runs at the fp32 vector rate
// (157 TF), so the contraction is MFMA-bound and the only way past that roof in fp32 is to multiply less.  Stock
// PyTorch-ROCm does the same for this network (MIOpen picks miopenSp3AsmConv_*_fp32_f2x3 / f3x2 Winograd kernels,
// profiles/archive/r02_torch_rocm_kernel_stats.csv), i.e. this is also the arithmetic the reference itself runs on a GPU.
// fp32 error of F(2x2,3x3) is of the direct convolution's order (transforms with coefficients 0, +-1, +-1/2 only):
// whole-UNet max error vs float64 1.35e-6 against 1.17e-6 for direct fp32 (tests/test_gpu_ops.py).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// i.e. 16 independent GEMMs (one per position (i,j) of the 4x4 transform domain)
//   M_ij[tile][n] = sum_c V_ij[tile][c] * U_ij[n][c],   tile = 2x2 output block, n = output channel, c = input channel.
//
// Workgroup = 8 waves (one per CU: 2 waves / SIMD): 64 Winograd tiles (16x16 output pixels of one image) x 64 output
// channels x all 16 positions (8 x 8 maps: four images per tile, NB4).  Wave w owns transform row i = w >> 1 and the two columns
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
// kernel.  Split-K over chunks writes output-domain slabs for k_splitk_reduce.  Train-mode dropout (block2's conv): the DROP
// instantiation masks the activated input in the staging step.
//
// U = G g G^T is derived from the OHWI weights by k_wino_weights whenever the weights change (plan-level "derived"
// buffer, 16/9 of the 3x3 weights' size); it is never part of a state dict.
#include <stdlib.h>

#include <algorithm>

#include "sr3_common.h"

#ifndef SR3_WINO_PRE
#define SR3_WINO_PRE 0
#endif
#ifdef SR3_SPLIT_NOSB
#define SR3_SB() do {} while (0)
#else
#define SR3_SB() __builtin_amdgcn_sched_barrier(0)
#endif

namespace sr3 {

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int WBN = 64;         // output channels per workgroup
constexpr int WCK = 16;         // input channels per chunk
constexpr int WRS = 20;         // LDS pixel stride (floats) of the raw halo: 16 channels + 4 pad
constexpr int WROW = 18 * WRS + 8;   // LDS row stride of the raw halo (18 pixels + 8 pad); every second ROW PAIR is shifted by
constexpr int WSHIFT = 4;            // 4 floats more: with this layout the transform's ds_read_b128 (lane = tile of a 4 x 8
                                     // tile block, two channel quads) hit 16 distinct bank quads per 16-lane group
constexpr int WNT = 512;        // threads (8 waves)
// Tile geometry.  NB4 = false: 64 Winograd tiles = 16 x 16 output pixels of ONE image, raw halo 18 x 18.  NB4 = true (8 x 8 maps):
// the 64 tiles are FOUR images of 8 x 8 pixels laid out as a 2 x 2 grid of 10 x 10 halos (each image keeps its own zero
// padding), i.e. a 20 x 20 raw block whose tile (ty, tx) starts 2 (ty >> 2) rows / 2 (tx >> 2) pixels further in.  The NB4
// form is split-K only (bias / FiLM / residual / statistics belong to the reduce kernel) with at most 16 chunks per split (the
// GroupNorm pairs of four images share the LDS block that holds one image's 64 chunks otherwise).
template <bool NB4> struct WGeo {
  static constexpr int TWp = NB4 ? 20 : 18;                 // raw halo pixels per row, and rows
  static constexpr int WROW = TWp * WRS + 8;                // LDS row stride
  static constexpr int SHIFT = NB4 ? 0 : WSHIFT;            // (the row-pair shift needs m blocks 8 rows apart)
  static constexpr int MROWS = NB4 ? 10 : 8;                // raw rows between the two 32-tile MFMA blocks
  static constexpr int WHP = TWp * TWp;                     // raw halo pixels
  static constexpr int WHI = (WHP * 4 + WNT - 1) / WNT;     // raw float4 items per thread (4 channel quads per pixel)
  static constexpr int RAW_F = TWp * WROW + 8;              // floats per raw buffer (two of them)
};
constexpr int WHP = WGeo<false>::WHP;
constexpr int WETS = 68;        // epilogue exchange: floats per channel row of a plane (64 tiles + 4 pad)
constexpr int WEPL = 2240;      // ... floats per plane: 32 rows x 68 + the 4-float shift of channels >= 16, padded to 16 bank quads x 35
constexpr int W_RAW_F = 18 * WROW + 8;                // floats per raw buffer (two of them)
constexpr int W_EXCH_F = 8 * 2 * WEPL;                // epilogue: 8 waves x 2 (q) planes
constexpr int W_MAX_CK = 64;                          // chunks of one workgroup's K range (1024 input channels; more: split-K)
constexpr int W_CST_F = 64 + W_MAX_CK * 2 * WCK;      // per-tile constants: bias + FiLM of 64 output channels, (scale, shift) pairs
                                                      // BEHIND the exchange block, two of them: the next tile's are written
                                                      // during the epilogue
static_assert(2 * W_RAW_F <= W_EXCH_F && W_RAW_F == WGeo<false>::RAW_F && 2 * WGeo<true>::RAW_F <= W_EXCH_F,
              "the raw tiles live inside the exchange block's footprint");
constexpr int W_SMEM = (W_EXCH_F + 2 * W_CST_F) * 4;   // 143,360 + 16,896 of the CU's 163,840 bytes
static_assert(W_SMEM <= 163840, "LDS");

// x * sigmoid(x) with the hardware exp2 (v_exp_f32 on x * log2 e) and the hardware reciprocal: ~1e-7 relative error on silu,
// three instructions instead of libm expf's twelve -- every staged element pays for this once per output-channel block.  The
// other kernels use libm expf (SR3_SILU); the difference is below what the parity tests resolve.
__device__ __forceinline__ float silu_w(float v) {
#ifdef SR3_EXACT_ACT
  return SR3_SILU(v);
#else
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
#endif
}
constexpr int WUS = 3 * 64 * 8;          // SPLIT: bf16 elements of one (position, n block) fragment group: 3 planes x 64 lanes x 8
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

// The same filters for the SPLIT kernel: every U value as three bf16 terms, laid out as the B operand of
// v_mfma_f32_32x32x16_bf16 (one MFMA = a whole 16-channel chunk):
//   ufrag_s[cout_blk][chunk][pos][nblk 2][plane 3 (h, m, l)][lane 64][8 bf16]
//   lane l holds U_pos[n = cout_blk*64 + nblk*32 + (l & 31)] for the 8 channels k = 0..7 of its half hq = l >> 5:
//   channel = chunk*16 + (k < 4 ? 4 hq + k : 8 + 4 hq + (k - 4))  -- the order in which the kernel's transform lanes hold
//   their channels (quad hq of either half chunk), so V needs no shuffle.  One thread per (n, chunk, hq).
__global__ __launch_bounds__(256) void k_wino_weights_split(const float* __restrict__ w, int Cout, int Cin, int nchunks,
                                                             int ncb, __bf16* __restrict__ ufrag) {
  const long total = (long)ncb * WBN * nchunks * 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int hq = (int)(idx & 1);
    const int chunk = (int)((idx >> 1) % nchunks);
    const int n = (int)((idx >> 1) / nchunks);
    f32x4 uu[2][16];                                  // [half chunk][position]
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int c = chunk * WCK + hf * 8 + hq * 4;
      f32x4 g[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        g[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (n < Cout && c < Cin) g[t] = *reinterpret_cast<const f32x4*>(w + ((size_t)n * 9 + t) * Cin + c);
      }
      f32x4 gg[4][3];
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) {
        const f32x4 g0 = g[0 * 3 + s_], g1 = g[1 * 3 + s_], g2 = g[2 * 3 + s_];
        gg[0][s_] = g0;
        gg[1][s_] = (g0 + g1 + g2) * 0.5f;
        gg[2][s_] = (g0 - g1 + g2) * 0.5f;
        gg[3][s_] = g2;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uu[hf][i * 4 + 0] = gg[i][0];
        uu[hf][i * 4 + 1] = (gg[i][0] + gg[i][1] + gg[i][2]) * 0.5f;
        uu[hf][i * 4 + 2] = (gg[i][0] - gg[i][1] + gg[i][2]) * 0.5f;
        uu[hf][i * 4 + 3] = gg[i][2];
      }
    }
    const int cb = n / WBN, nl = n - cb * WBN;
    const int nblk = nl >> 5;
    const int lane = (nl & 31) + 32 * hq;
    __bf16* base = ufrag + ((size_t)(cb * nchunks + chunk) * 16) * (2 * WUS) + (size_t)nblk * WUS + lane * 8;
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
      bf16x8 h, m, l;
      split3x8(uu[0][pos], uu[1][pos], h, m, l);
      __bf16* q = base + (size_t)pos * (2 * WUS);
      *reinterpret_cast<bf16x8*>(q) = h;
      *reinterpret_cast<bf16x8*>(q + 512) = m;
      *reinterpret_cast<bf16x8*>(q + 1024) = l;
    }
  }
}

size_t wino_weight_floats(int Cout, int Cin, bool split) {
  const size_t ncb = (Cout + WBN - 1) / WBN, nch = (Cin + WCK - 1) / WCK;
  return ncb * nch * 16 * (split ? (size_t)WUS : 1024);     // (split: 2 x WUS bf16 = WUS floats per position)
}

int wino_transform_weights(const float* w_ohwi, int Cout, int Cin, float* ufrag, hipStream_t st, bool split) {
  if ((Cin & 3) != 0) { set_error("wino: Cin %% 4 != 0"); return SR3_E_UNSUPPORTED; }
  const int ncb = (Cout + WBN - 1) / WBN, nch = (Cin + WCK - 1) / WCK;
  if (split) {
    const long total = (long)ncb * WBN * nch * 2;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_wino_weights_split, dim3(blocks), dim3(256), 0, st, w_ohwi, Cout, Cin, nch, ncb,
                       reinterpret_cast<__bf16*>(ufrag));
    SR3_LAUNCH_CHECK("k_wino_weights_split");
    return SR3_OK;
  }
  const long total = (long)ncb * WBN * nch * 4;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_wino_weights, dim3(blocks), dim3(256), 0, st, w_ohwi, Cout, Cin, nch, ncb, ufrag);
  SR3_LAUNCH_CHECK("k_wino_weights");
  return SR3_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// DBG (profiling only, env SR3_WINO_DBG, library built with -DSR3_WINO_ABLATIONS; 0 in production): 1 skip the MFMAs, 2 skip the
// GroupNorm / SiLU arithmetic of the staging step, 4 skip the input transforms, 8 skip the epilogue, 16 skip the U loads of the
// loop, 32 skip the raw staging of the loop, 64 write shader-clock stamps of every tile's phases (tools/wino_phases.py)
//
// Memory operations and the in-order vmcnt counter.  Measured on the round-3 phase timeline (profiles/archive/r03d_*): every
// `s_waitcnt vmcnt(0)` in a tile's prologue / epilogue costs 2-3 k cycles (all CUs hit their tile boundaries together), and
// the compiler falls back to vmcnt(0) whenever a load sits under a condition or in a loop of unknown length.  So outside the
// main loop every global load here is UNCONDITIONAL (absent operands are read from a valid dummy address and discarded by a
// select), issued in one fixed order, and consumed in that order:
//   epilogue:  [next tile's GroupNorm pairs + bias + FiLM]  [this tile's residual, both rounds]   ... round 0 ...
//              [next tile's raw chunks 0 and 1]             ... round 1 ...
//   prologue:  [U fragments of chunk 0]  stage chunks 0, 1  [raw chunk 2]
// DROP: train-mode dropout between the activation and the conv (nn.Dropout of Block, unet.py:86): the staged element with NHWC
// index i of the (single, non-upsampled) source is kept iff hash32(i * 0x9E3779B9 + seed) >= thresh and scaled by 1 / (1 - p),
// exactly as conv3x3_halo.hip does -- the mask applies to the activated input BEFORE the transform, so it fits the staging step.
// SPLIT: every fp32 operand as three bf16 terms (x = h + m + l) and each product as the six bf16 MFMA products
// hh + hm + mh + mm + hl + lh on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the dropped terms are <= 2^-24 of a product
// (fp32-class results, conv3x3_halo.hip's MODE 1 applied to the Winograd domain): U comes pre-split from the derived buffer
// (k_wino_weights_split), V is split in registers right after the transform.  Plan option `wino_split`; gated by tests.
template <int DBG, bool DROP, bool NB4, bool SPLIT>
__global__ __launch_bounds__(WNT, 1) void k_conv3x3_wino(const ConvParams p, const WinoGeom g,
                                                         const float* __restrict__ ufrag) {
  using GE = WGeo<NB4>;
  constexpr int WROW = GE::WROW, WHI = GE::WHI, WHP = GE::WHP, TWp = GE::TWp;
  extern __shared__ f32x4 smem_v[];
  float* smem = reinterpret_cast<float*>(smem_v);
  float* raw0 = smem;                             // [TWp rows][WROW] x 2 (double buffered), inside the exchange block's footprint
  float* raw1 = smem + GE::RAW_F;
  float* cst = smem + W_EXCH_F;                   // per-tile constants, double buffered by tile parity: [W_CST_F] x 2

  const int tid = threadIdx.x, lane = tid & 63;
  // wave-uniform on purpose (an SGPR): everything derived from it -- the transform row, the column pair, the signs -- is then
  // a scalar branch or a scalar select instead of per-lane v_cndmask work beside the MFMAs
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Cin = p.C0 + p.C1;
  const int H = p.Ho, W = p.Wo;
  // Persistent workgroups: workgroup b walks the tile list b, b + gridDim.x, ... (one 8-wave workgroup fits a CU; what a tile
  // needs before its main loop is fetched during the previous tile's epilogue, its stores drain under the next main loop).
  // XCD-aware order: consecutive workgroup ids go round-robin to the 8 XCDs; give each XCD one contiguous range of the
  // (cout block major) tile list so that the U fragments of a cout block stay inside one L2 (gridDim.x is a multiple of 8
  // whenever the tile count is, so a workgroup's tiles stay on its XCD's range).
  const int sp_tiles = g.tiles_w * g.tiles_h * g.nbt;            // nbt: batch tiles (B, or B / 4 for the four-image tile)
  const int ntiles = ((p.Cout + WBN - 1) / WBN) * sp_tiles;
  int cb = 0, tw_i = 0, th_i = 0, b0 = 0, h0 = 0, w0 = 0;
  auto decode_tile = [&](int v) {
    int bid = v;
    if ((ntiles & 7) == 0) bid = (bid & 7) * (ntiles >> 3) + (bid >> 3);
    // bid / sp_tiles by a host-side magic number (0: one tile per cout block, the quotient is bid itself)
    cb = g.sp_magic ? (int)(((unsigned long long)(unsigned)bid * g.sp_magic) >> 32) : bid;
    int sp = bid - cb * sp_tiles;
    if (g.pow2) {
      tw_i = sp & (g.tiles_w - 1);
      th_i = (sp >> g.log_tw) & (g.tiles_h - 1);
      b0 = sp >> (g.log_tw + g.log_th);
    } else {
      tw_i = sp % g.tiles_w;
      sp /= g.tiles_w;
      th_i = sp % g.tiles_h;
      b0 = sp / g.tiles_h;
    }
    h0 = th_i * 16; w0 = tw_i * 16;               // one image per tile
    if (NB4) { b0 *= 4; h0 = 0; w0 = 0; }         // ... or four whole 8 x 8 images: b0 is the first of them
  };

  const int nch = (Cin + WCK - 1) / WCK;
  const int cper = (nch + p.ksplit - 1) / p.ksplit;
  const int c_begin = blockIdx.y * cper;
  const int c_end = min(nch, c_begin + cper);
  const int nck = c_end - c_begin;                // 1 .. W_MAX_CK (host)
  const bool direct = !NB4 && p.ksplit == 1;      // (the four-image tile is split-K only: host)

  // ---- raw staging items of this thread: item j covers halo pixel (tid >> 2) + 128 j, channel quad tid & 3 ----
  // hinfo packs what is tile-independent: LDS float offset (bits 0..15), halo row (16..23), halo column (24..31); -1: no item
  const int kq = tid & 3, lrow = tid >> 2;
  int hinfo_r[WHI], hpix[WHI];
  auto hinfo_of = [&](int j, int t_) {
    const int hp = (t_ >> 2) + (WNT / 4) * j;
    const int hy = hp / TWp, hx = hp - hy * TWp;
    return hp < WHP ? (hy * WROW + ((hy >> 1) & 1) * GE::SHIFT + hx * WRS) | (hy << 16) | (hx << 24) : -1;
  };
#pragma unroll
  for (int j = 0; j < WHI; ++j) {
    hinfo_r[j] = hinfo_of(j, tid);
    hpix[j] = -1;
  }
  // SPLIT has no registers for these six values across its main loop: once a tile's exchange block is free (prologue) they are
  // parked in LDS, in the part of that block the raw tiles do not use -- [item][thread], every thread reads only what it wrote --
  // and re-read (three 8-byte reads) right before every staging step of the loop; the next tile's values are computed into the
  // registers again in the epilogue
  int* ptab = reinterpret_cast<int*>(smem + 2 * GE::RAW_F);       // [item][thread] of (hinfo, hpix)
  static_assert(!SPLIT || 2 * GE::RAW_F + 2 * WHI * WNT + 3 * WNT * 4 <= W_EXCH_F, "LDS tables of the SPLIT instantiation: raw tiles, parked staging items, position b's parked planes");
  auto hinfo = [&](int j) { return hinfo_r[j]; };
  auto pixel_of = [&](int j) {                    // source pixel of staging item j of the current tile (-1: zero padding)
    const int hj = hinfo_r[j];
    int hy = (hj >> 16) & 0xff, hx = (hj >> 24) & 0xff;
    int bi = b0;
    if (NB4) {                                      // halo block (hy / 10, hx / 10) of the 2 x 2 image grid
      const int iy = hy >= 10 ? 1 : 0, ix = hx >= 10 ? 1 : 0;
      bi += iy * 2 + ix; hy -= 10 * iy; hx -= 10 * ix;
    }
    const int ih = h0 + hy - 1, iw = w0 + hx - 1;
    const bool ok = hj >= 0 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    return ok ? (bi * p.Hs + (ih >> p.ups)) * p.Ws + (iw >> p.ups) : -1;
  };
  auto set_pixels = [&]() {
    if (SPLIT) {                                    // (recomputed per tile: not live across the main loop)
      int t_ = tid;
      asm volatile("" : "+v"(t_));
#pragma unroll
      for (int j = 0; j < WHI; ++j) hinfo_r[j] = hinfo_of(j, t_);
    }
#pragma unroll
    for (int j = 0; j < WHI; ++j) hpix[j] = pixel_of(j);
  };
  auto hpx = [&](int j) { return hpix[j]; };
  typedef int int2_t __attribute__((ext_vector_type(2)));
  auto park_items = [&]() {                        // SPLIT, prologue: registers -> LDS
    if (!SPLIT) return;
#pragma unroll
    for (int j = 0; j < WHI; ++j) reinterpret_cast<int2_t*>(ptab)[j * WNT + tid] = int2_t{hinfo_r[j], hpix[j]};
  };
  auto fetch_items = [&](int j0 = 0, int j1 = GE::WHI) {       // SPLIT, main loop: LDS -> registers, right before a staging step
    if (!SPLIT) return;
    int t_ = tid;
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int j = 0; j < WHI; ++j) {
      if (j < j0 || j >= j1) continue;
      const int2_t v = reinterpret_cast<const int2_t*>(ptab)[j * WNT + t_];
      hinfo_r[j] = v.x; hpix[j] = v.y;
    }
  };
  f32x4 rh[WHI];            // staging registers of the main loop (and of the tile's chunk 0)
  f32x4 rh2[WHI];           // ... of the tile's chunk 1: fetched during the previous tile's epilogue, idle in the main loop
  auto load_raw = [&](int chunk, f32x4 (&r)[WHI], int j0 = 0, int j1 = GE::WHI) {
    const int c = chunk * WCK + kq * 4;
    const int ce = c < Cin ? c : 0;
    const bool second = ce >= p.C0;
    const float* sp_ = second ? p.src1 : p.src0;
    const int sC = second ? p.C1 : p.C0;
    const int cs = second ? ce - p.C0 : ce;
#pragma unroll
    for (int j = 0; j < WHI; ++j) {
      if (j < j0 || j >= j1) continue;
      const int hp_ = hpx(j);
      const int off = hp_ >= 0 ? hp_ * sC + cs : 0;
      r[j] = *reinterpret_cast<const f32x4*>(sp_ + off);
    }
  };
  // GroupNorm (scale, shift) of the tile's image for the channels of this workgroup's chunk range live in LDS (cst + 64 of the
  // tile's parity): the staging step reads its 4 channels' pairs from there instead of keeping 8 registers live across a chunk
  [[maybe_unused]] auto load_pairs = [&](int chunk, const float* cs_, f32x4 (&ss)[2]) {       // (one-image tile) the staging step's pairs, read ahead
    ss[0] = f32x4{0.f, 0.f, 0.f, 0.f}; ss[1] = ss[0];
    if (p.act != 0) {
      int kq_ = kq;
      asm volatile("" : "+v"(kq_));
      const float* q = cs_ + 64 + (chunk - c_begin) * (2 * WCK) + kq_ * 8;
      ss[0] = *reinterpret_cast<const f32x4*>(q);
      ss[1] = *reinterpret_cast<const f32x4*>(q + 4);
    }
  };
  auto store_raw = [&](float* raw, int chunk, const f32x4 (&r)[WHI], const float* cs_, int j0 = 0, int j1 = GE::WHI,
                       const f32x4* pss = nullptr) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const bool hvalid = chunk * WCK + kq * 4 < Cin;
    f32x4 ssa = zero, ssb = zero;
    if (pss) { ssa = pss[0]; ssb = pss[1]; }
    else if (p.act != 0 && !NB4) {
      int kq_ = kq;
      asm volatile("" : "+v"(kq_));
      const float* q = cs_ + 64 + (chunk - c_begin) * (2 * WCK) + kq_ * 8;
      ssa = *reinterpret_cast<const f32x4*>(q);
      ssb = *reinterpret_cast<const f32x4*>(q + 4);
    }
#pragma unroll
    for (int j = 0; j < WHI; ++j) {
      if (j < j0 || j >= j1) continue;
      if (lrow + (WNT / 4) * j < WHP) {
        f32x4 v = r[j];
        if (NB4 && p.act != 0) {                    // the pairs of THIS item's image: [image][chunk of the split][16 x 2]
          const int hi_ = hinfo(j);
          const int img = (((hi_ >> 16) & 0xff) >= 10 ? 2 : 0) + (((hi_ >> 24) & 0xff) >= 10 ? 1 : 0);
          const float* q = cs_ + 64 + (img * nck + (chunk - c_begin)) * (2 * WCK) + kq * 8;
          ssa = *reinterpret_cast<const f32x4*>(q);
          ssb = *reinterpret_cast<const f32x4*>(q + 4);
        }
        if (p.act != 0 && !(DBG & 2)) {
          v.x = fmaf(v.x, ssa.x, ssa.y);
          v.y = fmaf(v.y, ssa.z, ssa.w);
          v.z = fmaf(v.z, ssb.x, ssb.y);
          v.w = fmaf(v.w, ssb.z, ssb.w);
          if (p.act == 2) { v.x = silu_w(v.x); v.y = silu_w(v.y); v.z = silu_w(v.z); v.w = silu_w(v.w); }
          if (DROP) {                                       // single source, no upsampling (host): linear NHWC index
            const unsigned i0 = (unsigned)(hpx(j) * p.C0 + chunk * WCK + kq * 4);
            v.x *= drop_mask(p.drop_seed, i0, p.drop_thresh, p.drop_scale);
            v.y *= drop_mask(p.drop_seed, i0 + 1, p.drop_thresh, p.drop_scale);
            v.z *= drop_mask(p.drop_seed, i0 + 2, p.drop_thresh, p.drop_scale);
            v.w *= drop_mask(p.drop_seed, i0 + 3, p.drop_thresh, p.drop_scale);
          }
        }
        v = (hvalid && hpx(j) >= 0) ? v : zero;
        int hi = hinfo(j);
        asm volatile("" : "+v"(hi));                // (the LDS address is derived here, not kept in a register of its own)
        *reinterpret_cast<f32x4*>(&raw[(hi & 0xffff) + kq * 4]) = v;
      }
    }
  };
  // Per-tile constants: bias + FiLM row of the tile's 64 output channels (cst[0..63]) and the (scale, shift) pairs of channels
  // [16 c_begin, 16 c_end) (cst[64..]).  Two unconditional loads + two of pairs per thread (nck <= 64: host), then LDS.
  const float* dummy = p.w;                          // valid memory for the loads of absent operands
  f32x4 creg[4];
  auto load_consts = [&]() {
    int t_ = tid;
    asm volatile("" : "+v"(t_));                     // (addresses recomputed per tile, not kept live across the main loop)
    const int n = cb * WBN + (t_ & 15) * 4;
    const int ne = n < p.Cout ? n : 0;
    creg[0] = *reinterpret_cast<const f32x4*>((direct && p.bias ? p.bias : dummy) + ne);
    creg[1] = *reinterpret_cast<const f32x4*>(direct && p.film ? p.film + (size_t)b0 * p.film_stride + ne : dummy);
    const float* q = p.act != 0 ? p.ss + (size_t)b0 * Cin * 2 : dummy;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      int e = t_ + WNT * k, img = 0;                             // 4 floats = 2 channels
      if (NB4) {                                                 // [image 0..3][nck * 8 float4]
        const int n8 = nck * (2 * WCK / 4);
        img = (e >= n8 ? 1 : 0) + (e >= 2 * n8 ? 1 : 0) + (e >= 3 * n8 ? 1 : 0);
        e -= img * n8;
        if (t_ + WNT * k >= 4 * n8) { e = 0; img = 0; }
      }
      const int ch = c_begin * WCK + e * 2;
      creg[2 + k] = *reinterpret_cast<const f32x4*>(q + (p.act != 0 && ch < Cin ? (size_t)img * Cin * 2 + (size_t)ch * 2 : 0));
    }
  };
  auto store_consts = [&](float* cs_) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    if (t_ < 16) {
      const int n = cb * WBN + t_ * 4;
      f32x4 v = zero;
      if (direct && n < p.Cout) v = (p.bias ? creg[0] : zero) + (p.film ? creg[1] : zero);
      *reinterpret_cast<f32x4*>(cs_ + t_ * 4) = v;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = t_ + WNT * k;
      const int n8 = nck * (2 * WCK) / 4;
      int ew = e;                                                // float4 index inside its image's block
      if (NB4) ew = e - ((e >= n8 ? 1 : 0) + (e >= 2 * n8 ? 1 : 0) + (e >= 3 * n8 ? 1 : 0)) * n8;
      const int ch = c_begin * WCK + ew * 2;
      if (e < (NB4 ? 4 : 1) * n8) *reinterpret_cast<f32x4*>(cs_ + 64 + e * 4) = (p.act != 0 && ch < Cin) ? creg[2 + k] : zero;
    }
  };

  // ---- this wave's transform row / columns ----
  const int wi = wave >> 1, wh = wave & 1;
  // rows of the 4x4 patch combined by B^T row wi:  t = d[ra] + sgn * d[rb]
  const int ra = (wi == 0) ? 0 : (wi == 2 ? 2 : 1);
  const int rb = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
  const float rsgn = (wi == 1) ? 1.f : -1.f;
  // Transform lanes: the 64 tiles are two blocks of 4 x 8 tiles (= the two 32-row MFMA blocks m); lane l works on tile
  // l & 31 of the block and on channel quad 2 kk + (l >> 5) of the half chunk kk -- exactly the A operand layout of
  // v_mfma_f32_32x32x2_f32 (row l & 31, k-half l >> 5), so the transformed values are MFMA operands as they are: V never
  // goes through LDS.  Float offsets of the two patch rows this wave combines (tile block m = 1 is 8 rows further down:
  // + 8 WROW, same shift parity -- an immediate offset of the read):
  const int tl = lane & 31, hq = lane >> 5;
  const int tyl = tl >> 3, tx = tl & 7;
  const int r0 = 2 * tyl;
  const int txo = 2 * tx + (NB4 ? 2 * (tx >> 2) : 0) + wh;          // (NB4: the right-hand images start 2 pixels further in)
  const int offa = (r0 + ra) * WROW + (((r0 + ra) >> 1) & 1) * GE::SHIFT + txo * WRS + hq * 4;
  const int offb = (r0 + rb) * WROW + (((r0 + rb) >> 1) & 1) * GE::SHIFT + txo * WRS + hq * 4;
  // the transform of one (m block, half chunk), split so that its six LDS reads can be in flight across an MFMA block:
  //   t_load issues the reads, t_finish does the row pass (3 FMAs) and the column pass (2 adds): va / vb = this wave's two
  //   positions, components = 4 consecutive k-steps
  auto t_load = [&](const float* rawbuf, int m, int kk, f32x4 (&da)[3], f32x4 (&db)[3]) {
    if (DBG & 4) return;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      da[s] = *reinterpret_cast<const f32x4*>(rawbuf + offa + (m * GE::MROWS * WROW + s * WRS + kk * 8));
      db[s] = *reinterpret_cast<const f32x4*>(rawbuf + offb + (m * GE::MROWS * WROW + s * WRS + kk * 8));
    }
  };
  auto t_finish = [&](const f32x4 (&da)[3], const f32x4 (&db)[3], f32x4& va, f32x4& vb) {
    if (DBG & 4) return;
    f32x4 t[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) t[s] = da[s] + db[s] * rsgn;
    // wh == 0: columns j = 0, 1 from t0, t1, t2:  V0 = t0 - t2, V1 = t1 + t2
    // wh == 1: columns j = 2, 3 from t1, t2, t3:  V2 = t2 - t1, V3 = t1 - t3     (t[] = t1, t2, t3)
    if (wh == 0) { va = t[0] - t[2]; vb = t[1] + t[2]; }
    else { va = t[1] - t[0]; vb = t[0] - t[2]; }
  };

  // ---- U fragments: [pj][nblk][kk] float4, straight from global in fragment-major order ----
  f32x4 u[2][2][2];
  const float* ubase = nullptr;                     // set per tile (cout block)
  auto load_u = [&](int chunk, int kk) {             // the four fragments of half a chunk (channels 8 kk .. 8 kk + 7)
    if ((DBG & 16) && chunk != c_begin) return;
    const float* q = ubase + (size_t)chunk * 16 * 1024;
#pragma unroll
    for (int pj = 0; pj < 2; ++pj)
#pragma unroll
      for (int n = 0; n < 2; ++n) u[pj][n][kk] = *reinterpret_cast<const f32x4*>(q + pj * 1024 + (n * 2 + kk) * 256);
  };

  // SPLIT: [pj][nblk][plane] bf16x8, one whole chunk per fragment; single-buffered, re-filled group by group (pj) right after
  // the group's last MFMA of the chunk
  bf16x8 us[2][2][3];
  const __bf16* ubase_s = nullptr;
  auto load_us = [&](int chunk, int pj) {
    if ((DBG & 16) && chunk != c_begin) return;
    // scalar base + 32-bit lane offset (the saddr form of the load): no 64-bit address pair lives across the loop
    const char* q = reinterpret_cast<const char*>(ubase_s + (size_t)chunk * 16 * (2 * WUS) + pj * (2 * WUS));
    unsigned l_ = (unsigned)lane;
    asm volatile("" : "+v"(l_));
    const unsigned vo = l_ * 16u;                  // zero-extended 32-bit lane offset: global_load ... v, s[base:base+1] offset:imm
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const char* qn = q + n * (WUS * 2);          // (its own scalar base: the immediate offsets stay below 4 KB)
      asm volatile("" : "+s"(qn));
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)           // (explicitly global: behind the laundering the pointer would be a flat one)
        us[pj][n][pl] = *(const __attribute__((address_space(1))) bf16x8*)(qn + (size_t)vo + pl * 1024);
    }
  };

  f32x16 acc[2][2][2];          // [pj][mblk][nblk]
  // SPLIT: the 12 MFMAs of one position (pj) of tile block m: product-major over the two n blocks (two independent
  // accumulators between dependent MFMAs), smallest terms first
  auto mfma_split = [&](int m, int pj, const bf16x8 (&v)[3]) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    if (DBG & 1) {               // keep the operands live, issue no MFMA
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) acc[pj][m][n][pl] += (float)v[pl][0] * (float)us[pj][n][pl][0];
      return;
    }
#ifdef SR3_SPLIT_PRIO_GROUPS
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int n = 0; n < 2; ++n)
        acc[pj][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[PA[q]], us[pj][n][PB[q]], acc[pj][m][n], 0, 0, 0);
#ifdef SR3_SPLIT_PRIO_GROUPS
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  auto mfma_unit = [&](int m, int kk, const f32x4& va, const f32x4& vb) {   // 16 MFMAs: tile block m, channels 8 kk .. 8 kk + 7
    if (DBG & 1) {               // keep the operands live, issue no MFMA
#pragma unroll
      for (int n = 0; n < 2; ++n) { acc[0][m][n][0] += va[0] * u[0][n][kk][0]; acc[1][m][n][0] += vb[0] * u[1][n][kk][0]; }
      return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        acc[0][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[q], u[0][n][kk][q], acc[0][m][n], 0, 0, 0);
        acc[1][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[q], u[1][n][kk][q], acc[1][m][n], 0, 0, 0);
      }
  };

  // DBG & 64: shader-clock stamps of the phases of every tile (lane 0 of every wave), 16 slots per (tile, wave), into
  // ConvParams::partial
  auto stamp = [&](int tile, int k) {
    if ((DBG & 64) && lane == 0 && p.partial)
      reinterpret_cast<unsigned long long*>(p.partial)[((size_t)tile * 8 + wave) * 16 + k] = __builtin_readcyclecounter();
  };

#ifdef SR3_WINO_SETPRIO
  // static priority for the second-dispatched half of the workgroup (the arbitration loser on every SIMD)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  // ---- first tile: constants, raw chunks 0 and 1 ------------------------------------------------------------------
  int vtile = blockIdx.x;
  int par = 0;                                       // parity of the tile: which half of the constants block it uses
  decode_tile(vtile);
  set_pixels();
  load_consts();
  const int c1 = min(c_begin + 1, c_end - 1), c2 = min(c_begin + 2, c_end - 1);   // (short K ranges re-fetch their last chunk:
  load_raw(c_begin, rh);                                                          //  the loads stay unconditional)
  load_raw(c1, rh2);
  store_consts(cst);
  __syncthreads();

  const int T = g.tiles_h * g.tiles_w;                             // statistics partials per image
  const bool stats = !NB4 && direct && p.ostat != nullptr;
  const bool has_res = direct && p.res0 != nullptr;
  const size_t Mtot = (size_t)p.B * H * W;
  float* dst = direct ? p.out : p.partial + (size_t)blockIdx.y * Mtot * p.Cout;
  if (DBG & 64) dst = p.out;

  for (;;) {
    // ================================ prologue ================================
    stamp(vtile, 0);
    const float* cs_ = cst + par * W_CST_F;
    {
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      ubase = ufrag + (size_t)cb * nch * 16 * 1024 + (size_t)(wi * 4 + wh * 2) * 1024 + l_ * 4;
      ubase_s = reinterpret_cast<const __bf16*>(ufrag) + (size_t)cb * nch * 16 * (2 * WUS) + (size_t)(wi * 4 + wh * 2) * (2 * WUS);     // (wave-uniform)
    }
    if (SPLIT) {
      load_us(c_begin, 0);
      load_us(c_begin, 1);
    } else {
      load_u(c_begin, 0);                             // L2 hits (every tile of the cout block reads them): back before the
      load_u(c_begin, 1);                             // two staging steps below are done
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;
    park_items();
    store_raw(raw0, c_begin, rh, cs_);
    if (nck > 1) store_raw(raw1, c_begin + 1, rh2, cs_);
    load_raw(c2, rh);
    __syncthreads();
    stamp(vtile, 1);

    // ================================ main loop ================================
    // A stream of units (m block, half chunk), software pipelined one unit deep.  Chunk i's raw tile lives in raw[i & 1].
    // Unit order per chunk: (m0, kk0) (m1, kk0) (m0, kk1) (m1, kk1); the six LDS reads of a unit are issued before the
    // PREVIOUS unit's MFMA block and consumed after it, its operands then stay in registers (va, vb) for its own block:
    //   reads (m1,kk0)  | MFMA (m0,kk0) | finish (m1,kk0)
    //   reads (m0,kk1)  | MFMA (m1,kk0) | finish (m0,kk1)        U fragments kk0 of chunk i + 1
    //   reads (m1,kk1)  | MFMA (m0,kk1) | finish (m1,kk1)
    //   barrier: raw[i & 1] is fully consumed -> stage chunk i + 2 into it; chunk i + 1's tile is visible
    //   reads (m0,kk0) of chunk i + 1 | MFMA (m1,kk1) | finish    U fragments kk1 of chunk i + 1
    if constexpr (SPLIT) {
      // Units are whole tile blocks (m) of a chunk: K = 16 is ONE bf16 MFMA per (position, n block, product).  Per unit two MFMA
      // groups of 12 (position a, position b); behind each group the transform of half of the NEXT unit (kk = 0 / kk = 1: six LDS
      // reads, row + column pass), behind the second also the 3 x bf16 split of the next unit's operands.  The U fragments of
      // the next chunk are fetched per position right after that position's last MFMA of the chunk: one buffer, refilled >= 1 k
      // cycles ahead of its next use.
      //   MFMA (m0, a) x 12 | reads kk0 (m1), finish ; fetch V_b (m0) from LDS
      //   MFMA (m0, b) x 12 | reads kk1 (m1), finish ; park V_b (m1) in LDS ; split -> V_a (m1)
      //   barrier ; stage chunk i + 2
      //   MFMA (m1, a) x 12 | U pos a of chunk i + 1 ; reads kk0 (m0 of chunk i + 1), finish ; fetch V_b (m1)
      //   MFMA (m1, b) x 12 | U pos b of chunk i + 1 ; reads kk1, finish ; park V_b (m0, i + 1) ; split -> V_a (m0, i + 1)
      constexpr int PRE = DROP ? 0 : SR3_WINO_PRE;       // (the dropout instantiation has no registers for the early reads)
      bf16x8 vsa[3], vsb[3];
      f32x4 va0 = {0.f, 0.f, 0.f, 0.f}, vb0 = va0, va1 = va0, vb1 = va0;
      auto sp3 = [&](const f32x4& lo, const f32x4& hi, bf16x8& h, bf16x8& m, bf16x8& l) {
        if (DBG & 128) {           // ablation: one conversion per value, no residuals
#pragma unroll
          for (int e = 0; e < 8; ++e) h[e] = (__bf16)(e < 4 ? lo[e] : hi[e - 4]);
          m = h; l = h;
          return;
        }
        split3x8(lo, hi, h, m, l);
      };
      {
        f32x4 da[3], db[3];
        t_load(raw0, 0, 0, da, db);
        t_finish(da, db, va0, vb0);
        t_load(raw0, 0, 1, da, db);
        t_finish(da, db, va1, vb1);
        sp3(va0, va1, vsa[0], vsa[1], vsa[2]);
      }
      // Registers: this instantiation has none to spare (128 accumulators + 48 of U + the operands), so (a) no LDS read is in
      // flight across an MFMA group -- the six reads of a half-unit are issued right behind a group (the partner wave of the
      // SIMD owns the matrix pipe meanwhile), and (b) position b's split planes wait in LDS ([plane][thread], 16 bytes each, in
      // the free part of the exchange block) from the moment they are built until their own MFMA group: 12 registers across group a.
      bf16x8* vpark = reinterpret_cast<bf16x8*>(ptab + 2 * WHI * WNT);
      auto park_b = [&]() {
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        bf16x8 h, m, l;
        sp3(vb0, vb1, h, m, l);
        if (DBG & 256) { vsb[0] = h; vsb[1] = m; vsb[2] = l; return; }      // ablation: no LDS round trip
        vpark[t_] = h; vpark[WNT + t_] = m; vpark[2 * WNT + t_] = l;
      };
      auto fetch_b = [&]() {
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        if (DBG & 256) return;
        vsb[0] = vpark[t_]; vsb[1] = vpark[WNT + t_]; vsb[2] = vpark[2 * WNT + t_];
      };
      park_b();
#if !defined(SR3_SPLIT_ROT) || SR3_SPLIT_ROT == 1
      const bool late = wave >= 4;                    // (waves w and w + 4 share SIMD w & 3)
#elif SR3_SPLIT_ROT == 2
      const bool late = (wave & 1) != 0;
#else
      const bool late = false;
#endif
      for (int i = 0; i < nck; ++i) {
        float* rcur = (i & 1) ? raw1 : raw0;
        const float* rnext = (i & 1) ? raw0 : raw1;
        const bool more = i + 1 < nck;
        f32x4 da[3], db[3];
        // SR3_WINO_PRE & 1 (round 5): the six row reads of the next half-unit are issued BEFORE the MFMA group and consumed behind
        // it, so their LDS latency runs under the group (plain LDS traffic hides beside an MFMA: tools/mfma_fillers.hip); the
        // first half of the iteration has the registers for it (0 spills), the second half -- staging loads in flight -- has not
        if (PRE & 1) t_load(rcur, 1, 0, da, db);
        SR3_SB();
        mfma_split(0, 0, vsa);
        SR3_SB();
        if (!(PRE & 1)) t_load(rcur, 1, 0, da, db);
        t_finish(da, db, va0, vb0);
        if (!(PRE & 1)) SR3_SB();
        fetch_b();
        if (PRE & 1) t_load(rcur, 1, 1, da, db);
        SR3_SB();
        mfma_split(0, 1, vsb);
        SR3_SB();
        if (!(PRE & 1)) t_load(rcur, 1, 1, da, db);
        t_finish(da, db, va1, vb1);
        SR3_SB();
        park_b();
        SR3_SB();
        sp3(va0, va1, vsa[0], vsa[1], vsa[2]);
        // The two waves of a SIMD run this loop in lock step (same code, one barrier per chunk), so left alone both sit in their
        // MFMA groups together and in their transform / split sections together, and neither the matrix pipe nor the VALU is
        // ever busy while the other is.  `late` waves (one of each SIMD's pair) issue group (m1, a) BEFORE the barrier, the others
        // after it: from then on one wave of the pair is in an MFMA group while the other is in a VALU section.
        SR3_SB();
        if (late) {
          mfma_split(1, 0, vsa);
          if (more) load_us(c_begin + i + 1, 0);
        }
        SR3_SB();
        __syncthreads();
#ifdef SR3_SPLIT_SKEW
        if (wave >= 4) __builtin_amdgcn_s_sleep(SR3_SPLIT_SKEW);      // (64 cycles per unit) de-phase the two waves of a SIMD
#endif
        if (i + 2 < nck && !(DBG & 32)) {
          fetch_items();
          store_raw(rcur, c_begin + i + 2, rh, cs_);
          if (i + 3 < nck) load_raw(c_begin + i + 3, rh);
        }
        SR3_SB();
        if (!late) {
          mfma_split(1, 0, vsa);
          if (more) load_us(c_begin + i + 1, 0);
        }
        SR3_SB();
        if (more) {
          t_load(rnext, 0, 0, da, db);
          t_finish(da, db, va0, vb0);
        }
        SR3_SB();
        fetch_b();
        if ((PRE & 2) && more) t_load(rnext, 0, 1, da, db);
        SR3_SB();
        mfma_split(1, 1, vsb);
        SR3_SB();
        if (more) {
          load_us(c_begin + i + 1, 1);
          if (!(PRE & 2)) t_load(rnext, 0, 1, da, db);
          t_finish(da, db, va1, vb1);
          SR3_SB();
          park_b();
          SR3_SB();
          sp3(va0, va1, vsa[0], vsa[1], vsa[2]);
        }
      }
    } else {
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      {
        f32x4 da[3], db[3];
        t_load(raw0, 0, 0, da, db);
        t_finish(da, db, va, vb);
      }
      for (int i = 0; i < nck; ++i) {
        float* rcur = (i & 1) ? raw1 : raw0;
        const float* rnext = (i & 1) ? raw0 : raw1;
        const bool more = i + 1 < nck;
        f32x4 da[3], db[3];
        // (sched_barrier: keep the next unit's LDS reads ahead of the MFMA block and its arithmetic behind it -- left
        // alone the scheduler sinks the reads below the MFMAs and waits on them at once)
        t_load(rcur, 1, 0, da, db);
        __builtin_amdgcn_sched_barrier(0);
        mfma_unit(0, 0, va, vb);
        __builtin_amdgcn_sched_barrier(0);
        t_finish(da, db, va, vb);
        t_load(rcur, 0, 1, da, db);
        __builtin_amdgcn_sched_barrier(0);
        mfma_unit(1, 0, va, vb);
        __builtin_amdgcn_sched_barrier(0);
        t_finish(da, db, va, vb);
        if (more) load_u(c_begin + i + 1, 0);
        t_load(rcur, 1, 1, da, db);
        __builtin_amdgcn_sched_barrier(0);
        mfma_unit(0, 1, va, vb);
        __builtin_amdgcn_sched_barrier(0);
        t_finish(da, db, va, vb);
        __syncthreads();
        if (i + 2 < nck && !(DBG & 32)) {
          store_raw(rcur, c_begin + i + 2, rh, cs_);
          if (i + 3 < nck) load_raw(c_begin + i + 3, rh);
        }
        if (more) t_load(rnext, 0, 0, da, db);
        __builtin_amdgcn_sched_barrier(0);
        mfma_unit(1, 1, va, vb);
        __builtin_amdgcn_sched_barrier(0);
        if (more) { t_finish(da, db, va, vb); load_u(c_begin + i + 1, 1); }
      }
    }

    // ================================ epilogue ================================
    // fold the two columns of this wave with A (A^T = [1 1 1 0; 0 1 -1 -1]):  P_q = sum_j M_ij A[j][q]
    //   wh == 0 (j = 0, 1): P_0 = M0 + M1, P_1 = M1          wh == 1 (j = 2, 3): P_0 = M2, P_1 = -M2 - M3
    // then Y[p][q] = sum_i A^T[p][i] (P_q(i,0) + P_q(i,1)) through LDS in a FIXED order (bitwise reproducible), one 32-channel
    // block per round.  Exchange layout: plane (wave, q) = [32 channels][64 tiles + 4 pad], channels >= 16 shifted by 4 floats
    // more, so that both the 16-byte writes (8 consecutive channels per lane group) and the 16-byte reads are bank-conflict
    // free; thread -> (sub-pixel p q, 4 consecutive tiles, 4 consecutive channels): 24 reads of 4 tiles each, a 4 x 4
    // register transpose, 4 NHWC stores of 16 bytes (8 adjacent lanes = 128 contiguous bytes).
    // Lane roles: thread -> (sub-pixel p q, 4 consecutive tiles, 4 consecutive channels).  The lane id is laundered through an
    // empty asm so that this index arithmetic is redone per tile instead of being hoisted out of the tile loop, where it would
    // occupy registers -- or scratch, whose reloads drain the memory queue -- across the main loop.
    int elane = lane;
    asm volatile("" : "+v"(elane));
    const int etid = wave * 64 + elane;
    const int nq = elane & 7, fq = (elane >> 4) & 1, fp = wave >> 2;
    const int tq = ((elane >> 5) & 1) | (((elane >> 3) & 1) << 1) | ((wave & 3) << 2);   // tiles 4 tq .. 4 tq + 3 (half a tile row)
    const int ety = tq >> 1, etx0 = (tq & 1) * 4;                                       // tile row, first tile column
    const int e_cb = cb, e_b0 = b0, e_tix = th_i * g.tiles_w + tw_i, e_vtile = vtile;
    const size_t pix0 = NB4 ? ((size_t)(b0 + (ety >> 2) * 2 + (tq & 1)) * H + (2 * (ety & 3) + fp)) * W + fq
                            : ((size_t)b0 * H + (h0 + 2 * ety + fp)) * W + (w0 + 2 * etx0 + fq);   // tile k of the four: + 2 k pixels
    stamp(e_vtile, 2);
    // the next tile (the last one re-fetches itself: the loads stay unconditional)
    const bool has_next = vtile + (int)gridDim.x < ntiles;
    if (has_next) vtile += gridDim.x;
    // this tile's residual, both rounds: 8 loads (absent: a valid dummy address, discarded below)
    f32x4 addv[2][4];
#pragma unroll
    for (int nblk = 0; nblk < 2; ++nblk) {
      const int n = e_cb * WBN + nblk * 32 + nq * 4;
      const int ne = n < p.Cout ? n : 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t pix = pix0 + 2 * k;
        const float* rp = dst + pix * p.Cout + ne;
        if (has_res) rp = (ne < p.RC0) ? p.res0 + pix * p.RC0 + ne : p.res1 + pix * p.RC1 + (ne - p.RC0);
        addv[nblk][k] = *reinterpret_cast<const f32x4*>(rp);
      }
    }
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
      p.out[(size_t)e_vtile * WNT + etid] = s + addv[0][0][0] + addv[1][3][3];
      __syncthreads();
      decode_tile(vtile);
      load_consts();
      store_consts(cst + (par ^ 1) * W_CST_F);
      set_pixels();
      load_raw(c_begin, rh);
      load_raw(c1, rh2);
      __syncthreads();
      par ^= 1;
      if (!has_next) break;
      continue;
    }
    // bias + FiLM of this thread's two channel quads (LDS: written a tile ago)
    f32x4 base[2];
#pragma unroll
    for (int nblk = 0; nblk < 2; ++nblk) base[nblk] = *reinterpret_cast<const f32x4*>(cs_ + nblk * 32 + nq * 4);
    stamp(e_vtile, 3);
    __syncthreads();                                                 // the raw tiles are dead: the exchange block reuses LDS
    stamp(e_vtile, 4);
    float* exch = smem;                                              // [8 waves][2 q][WEPL]
    double* part = reinterpret_cast<double*>(smem);                  // statistics parking (after the last round)
    double s1[2][4], s2[2][4];                                       // this thread's channel quad of either 32-channel block
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int k = 0; k < 4; ++k) { s1[a][k] = 0.0; s2[a][k] = 0.0; }
    const int wn = elane & 31;                                       // channel this lane's accumulator column belongs to
    float* wbase = exch + (wave * 2) * WEPL + wn * WETS + (wn >> 4) * 4 + 4 * (elane >> 5);
#pragma unroll
    for (int nblk = 0; nblk < 2; ++nblk) {
      const int n = e_cb * WBN + nblk * 32 + nq * 4;
      const bool nok = n < p.Cout;
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
        // D layout: reg r of lane l -> tile (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the block, channel l & 31
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          *reinterpret_cast<f32x4*>(wbase + mblk * 32 + 8 * k) = f32x4{p0[4 * k], p0[4 * k + 1], p0[4 * k + 2], p0[4 * k + 3]};
          *reinterpret_cast<f32x4*>(wbase + WEPL + mblk * 32 + 8 * k) = f32x4{p1[4 * k], p1[4 * k + 1], p1[4 * k + 2], p1[4 * k + 3]};
        }
      }
      stamp(e_vtile, 5 + 4 * nblk);
      if (nblk == 0) {
        // half of the accumulators are dead: the next tile's constants (4 loads) and raw chunks 0 and 1 are put in flight
        decode_tile(vtile);
        load_consts();
        set_pixels();
        load_raw(c_begin, rh);
        load_raw(c1, rh2);
      }
      stamp(e_vtile, 6 + 4 * nblk);
      __syncthreads();
      stamp(e_vtile, 7 + 4 * nblk);
      {
        // Y[p][q] = sum_i A^T[p][i] (P_q(i, 0) + P_q(i, 1)), rows in a fixed order: p = 0: (r0 + r1) + r2, p = 1: (r1 - r2) - r3
        f32x4 y[4];                                                  // [channel j] over the 4 tiles
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nn = nq * 4 + j;
          const float* rb_ = exch + nn * WETS + (nn >> 4) * 4 + 4 * tq;
          auto rd = [&](int i) {
            return *reinterpret_cast<const f32x4*>(rb_ + ((i * 2 + 0) * 2 + fq) * WEPL) +
                   *reinterpret_cast<const f32x4*>(rb_ + ((i * 2 + 1) * 2 + fq) * WEPL);
          };
          if (fp == 0) { const f32x4 r0_ = rd(0), r1_ = rd(1), r2_ = rd(2); y[j] = (r0_ + r1_) + r2_; }
          else { const f32x4 r1_ = rd(1), r2_ = rd(2), r3_ = rd(3); y[j] = (r1_ - r2_) - r3_; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f32x4 v = f32x4{y[0][k], y[1][k], y[2][k], y[3][k]};
          if (direct) {
            v += base[nblk];
            if (has_res) v += addv[nblk][k];
            if (stats) {
#pragma unroll
              for (int c = 0; c < 4; ++c) { const double dv = (double)v[c]; s1[nblk][c] += dv; s2[nblk][c] += dv * dv; }
            }
          }
          if (nok) *reinterpret_cast<f32x4*>(dst + (pix0 + 2 * k) * p.Cout + n) = v;
        }
      }
      if (nblk == 0) store_consts(cst + (par ^ 1) * W_CST_F);      // ... and the constants go to LDS (the other parity)
      stamp(e_vtile, 8 + 4 * nblk);
      __syncthreads();                                  // every read of the exchange block is complete
    }
    stamp(e_vtile, 13);
    if (stats) {
      // Per-channel sums of this tile's outputs (one image per tile), in a fixed order, as a two-level tree over LDS: every
      // thread parks its 16 partial sums e = [sum | sum of squares][32-channel block][channel of the quad] as part[e][thread]
      // (rows of 520 doubles: consecutive lanes write consecutive doubles, and the 4 rows a 32-lane group reads below sit 16
      // banks apart -- the thread-major layout cost 8 k cycles of bank conflicts per tile); 512 threads each add 16 of the 64
      // partials that share a channel quad (threads nq, nq + 8, ...), 128 threads add the 4 results.
      constexpr int PR = 520;
#pragma unroll
      for (int nb2 = 0; nb2 < 2; ++nb2)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          part[(nb2 * 4 + k) * PR + etid] = s1[nb2][k];
          part[(8 + nb2 * 4 + k) * PR + etid] = s2[nb2][k];
        }
      __syncthreads();
      {
        const int cq = etid & 7, e = (etid >> 3) & 15, grp = etid >> 7;
        double a = 0.0;
#pragma unroll
        for (int s = 0; s < 16; ++s) a += part[e * PR + (grp * 16 + s) * 8 + cq];
        part[16 * PR + grp * 128 + e * 8 + cq] = a;
      }
      __syncthreads();
      stamp(e_vtile, 14);
      if (etid < 128) {
        const int cq = etid & 7, e = etid >> 3;                   // e = which * 8 + nb2 * 4 + k
        double a = part[16 * PR + etid];
#pragma unroll
        for (int grp = 1; grp < 4; ++grp) a += part[16 * PR + grp * 128 + etid];
        const int which = e >> 3, c = ((e >> 2) & 1) * 32 + cq * 4 + (e & 3);
        const int nn = e_cb * WBN + c;
        if (nn < p.Cout) p.ostat[(((size_t)e_b0 * T + e_tix) * p.Cout + nn) * 2 + which] = a;
      }
    }
    __syncthreads();                                    // LDS (exchange block / parked sums) is free for the next tile
    stamp(e_vtile, 15);
    par ^= 1;
    if (!has_next) break;
  }   // tile loop
}

// ---- host -----------------------------------------------------------------------------------------------------
namespace {
inline int ilog2x(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
}  // namespace

bool wino_geometry(const ConvParams& p, WinoGeom* g) {
  if (p.ksize != 3 || p.stride != 1) return false;
  const int H = p.Ho, W = p.Wo;
  if (H != (p.Hs << p.ups) || W != (p.Ws << p.ups)) return false;
  // four whole 8 x 8 images per workgroup tile (split-K only: at least two 16-channel chunks)
  const bool nb4 = H == 8 && W == 8 && p.ups == 0 && (p.B % 4) == 0 && p.C0 + p.C1 > WCK;
  // wino_split == 2: the 8 x 16 pixel tile of conv3x3_wino2.hip (two four-wave workgroups per CU)
  const bool w2 = p.wino_split == 2;
  if (w2 && !wino2_fits(p)) return false;
  if (!w2 && !nb4 && (W < 16 || (W % 16) != 0 || (H % 16) != 0)) return false;
  g->TH = (nb4 || w2) ? 8 : 16; g->TW = nb4 ? 8 : 16; g->NB = nb4 ? 4 : 1;
  g->twt = 8; g->log_twt = 3;
  g->tpi = nb4 ? 16 : (w2 ? 32 : 64); g->log_tpi = nb4 ? 4 : (w2 ? 5 : 6);
  g->tiles_w = nb4 ? 1 : W / 16; g->tiles_h = nb4 ? 1 : H / g->TH;
  g->nbt = nb4 ? p.B / 4 : p.B;
  g->HPI = nb4 ? 100 : (w2 ? 180 : WHP);
  g->HP = nb4 ? 400 : (w2 ? 180 : WHP);
  // tile decode of the persistent kernel: shifts when the tile grid is a power of two, a magic number for / sp_tiles
  g->log_tw = ilog2x(g->tiles_w); g->log_th = ilog2x(g->tiles_h);
  g->pow2 = ((1 << g->log_tw) == g->tiles_w && (1 << g->log_th) == g->tiles_h) ? 1 : 0;
  const unsigned long long spt = (unsigned long long)g->tiles_w * g->tiles_h * g->nbt;
  if (spt == 0 || spt * spt * ((p.Cout + WBN - 1) / WBN) >= (1ull << 32)) return false;    // (keeps the multiply-high exact)
  g->sp_magic = spt == 1 ? 0u : (unsigned)(((1ull << 32) + spt - 1) / spt);
  return true;
}
int wino_stats_slices(const WinoGeom& g) { return g.NB == 1 ? g.tiles_h * g.tiles_w : 0; }   // (four-image tile: split-K only)
int wino_max_chunks_per_split(const WinoGeom& g) { return g.NB == 1 ? W_MAX_CK : 16; }
long wino_workgroups(const ConvParams& p, const WinoGeom& g) {
  return (long)((p.Cout + WBN - 1) / WBN) * g.tiles_w * g.tiles_h * g.nbt;
}
int wino_chunks(const ConvParams& p) { return (p.C0 + p.C1 + WCK - 1) / WCK; }

int conv3x3_wino_forward(const ConvParams& p, const float* ufrag, hipStream_t st) {
  WinoGeom g;
  if (!wino_geometry(p, &g)) { set_error("conv: the Winograd kernel does not fit this problem"); return SR3_E_UNSUPPORTED; }
  if (!ufrag) { set_error("conv: Winograd needs the transformed weights"); return SR3_E_BADARG; }
  if (p.x2_w) { set_error("conv: the Winograd kernel has no fused 1x1 segment"); return SR3_E_UNSUPPORTED; }
  if (p.drop_thresh != 0 && (p.C1 != 0 || p.ups != 0 || p.act == 0)) { set_error("conv: dropout needs a single-source, non-upsampled, activated input"); return SR3_E_UNSUPPORTED; }
  const int nch = wino_chunks(p);
  if ((nch + p.ksplit - 1) / p.ksplit > wino_max_chunks_per_split(g)) { set_error("conv: the Winograd kernel takes at most %d input channels per K split here (%d chunks over %d splits)", wino_max_chunks_per_split(g) * WCK, nch, p.ksplit); return SR3_E_UNSUPPORTED; }
  if (g.NB != 1 && p.ksplit < 2) { set_error("conv: the four-image Winograd tile (8 x 8 maps) is split-K only"); return SR3_E_UNSUPPORTED; }
  if (p.ksplit > 1 && (long)(p.ksplit - 1) * ((nch + p.ksplit - 1) / p.ksplit) >= nch) { set_error("conv: ksplit %d leaves an empty split over %d chunks", p.ksplit, nch); return SR3_E_BADARG; }
  // persistent workgroups: one per CU (8 waves, 157 KB of LDS), each walking its share of the tile list
  static const int n_cu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n & ~7;               // a multiple of 8 keeps every workgroup's tiles on one XCD's range of the list
  }();
  const long ntiles = wino_workgroups(p, g);
  const char* np = getenv("SR3_WINO_NONPERSISTENT");
  dim3 grid((unsigned)((np && np[0] == '1') ? ntiles : std::min<long>(ntiles, n_cu > 0 ? n_cu : 256)), p.ksplit);
  static const int dbg_env = [] { const char* e = getenv("SR3_WINO_DBG"); return e ? atoi(e) : 0; }();
  const int dbg = (g.NB != 1 || p.drop_thresh != 0) ? 0 : dbg_env;       // (the ablations cover the one-image tile without dropout)
#define SR3_WINO_LAUNCH4(D, DR, N4, SP)                                                                               \
  {                                                                                                                   \
    static std::atomic<uint64_t> done{0};                                                                             \
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(k_conv3x3_wino<D, DR, N4, SP>), W_SMEM, done)) return rc; \
    hipLaunchKernelGGL((k_conv3x3_wino<D, DR, N4, SP>), grid, dim3(WNT), W_SMEM, st, p, g, ufrag);                    \
  }
#define SR3_WINO_LAUNCH3(D, DR, N4) SR3_WINO_LAUNCH4(D, DR, N4, false)
#define SR3_WINO_LAUNCH2(D, DR)                                                                                       \
  if (g.NB == 1) SR3_WINO_LAUNCH3(D, DR, false) else SR3_WINO_LAUNCH3(D, DR, true)
#define SR3_WINO_LAUNCH(D)                                                                                            \
  if (g.NB != 1) { set_error("conv: the Winograd ablations cover the one-image tile only"); return SR3_E_BADARG; }    \
  SR3_WINO_LAUNCH3(D, false, false)
  if (p.drop_thresh != 0 && dbg != 0) { set_error("conv: the Winograd ablations have no dropout form"); return SR3_E_BADARG; }
  if (p.wino_split == 2) return conv3x3_wino2_forward(p, g, ufrag, st);
  if (p.wino_split && g.NB != 1 && p.drop_thresh != 0) {
    set_error("conv: the split-bf16 four-image Winograd tile has no dropout form");
    return SR3_E_UNSUPPORTED;
  }
  switch (dbg) {
    case 0:
      if (p.wino_split && p.drop_thresh != 0) { SR3_WINO_LAUNCH4(0, true, false, true) }
      else if (p.wino_split && g.NB != 1) { SR3_WINO_LAUNCH4(0, false, true, true) }
      else if (p.wino_split) { SR3_WINO_LAUNCH4(0, false, false, true) }
      else if (p.drop_thresh != 0) { SR3_WINO_LAUNCH2(0, true) } else { SR3_WINO_LAUNCH2(0, false) }
      break;
#ifdef SR3_WINO_ABLATIONS
    case 1: if (p.wino_split) { SR3_WINO_LAUNCH4(1, false, false, true) } else { SR3_WINO_LAUNCH(1) } break;
    case 2: { SR3_WINO_LAUNCH(2) } break;
    case 4: if (p.wino_split) { SR3_WINO_LAUNCH4(4, false, false, true) } else { SR3_WINO_LAUNCH(4) } break;
    case 8: if (p.wino_split) { SR3_WINO_LAUNCH4(8, false, false, true) } else { SR3_WINO_LAUNCH(8) } break;
    case 16: if (p.wino_split) { SR3_WINO_LAUNCH4(16, false, false, true) } else { SR3_WINO_LAUNCH(16) } break;
    case 32: if (p.wino_split) { SR3_WINO_LAUNCH4(32, false, false, true) } else { SR3_WINO_LAUNCH(32) } break;
    case 38: { SR3_WINO_LAUNCH(38) } break;       // MFMA + U + epilogue only
    case 46: { SR3_WINO_LAUNCH(46) } break;       // MFMA + U only
    case 62: { SR3_WINO_LAUNCH(62) } break;       // bare MFMA loop
    case 64: { SR3_WINO_LAUNCH(64) } break;       // phase time stamps
#define SR3_WINO_SPLIT_CASE(D) case D: if (p.wino_split) { SR3_WINO_LAUNCH4(D, false, false, true) } else { SR3_WINO_LAUNCH(D) } break;
    SR3_WINO_SPLIT_CASE(128)                      // SPLIT: no residual arithmetic in the 3 x bf16 split
    SR3_WINO_SPLIT_CASE(256)                      // SPLIT: position b's planes stay in registers (no LDS round trip)
    SR3_WINO_SPLIT_CASE(444)                      // SPLIT: bare MFMA loop + epilogue (4 + 16 + 32 + 128 + 256)
    SR3_WINO_SPLIT_CASE(452)                      // SPLIT: bare MFMA loop, no epilogue
    SR3_WINO_SPLIT_CASE(445)                      // SPLIT: nothing but the prologue / epilogue
#endif
    default: set_error("conv: SR3_WINO_DBG=%d is not built (compile with -DSR3_WINO_ABLATIONS)", dbg); return SR3_E_BADARG;
  }
#undef SR3_WINO_LAUNCH
#undef SR3_WINO_LAUNCH2
#undef SR3_WINO_LAUNCH3
#undef SR3_WINO_LAUNCH4
  SR3_LAUNCH_CHECK("k_conv3x3_wino");
  return SR3_OK;
}

}  // namespace sr3
