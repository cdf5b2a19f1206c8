// Winograd F(2x2, 3x3) on the bf16 MFMA with 3-way split operands, FOUR waves of 512 registers (one per SIMD).
//
// EXPERIMENT, off by default (plan option wino4 / tile_cfg 13): parity-green, and no faster than the 8-wave SPLIT instantiation it was
// meant to replace (8.01-8.4 vs 8.08 ms per step, profiles/r04e_wino4_ablations.txt).  The premise below -- that interleaving the
// staging / transform VALU work with the MFMAs inside one wave hides it -- does not hold on gfx950: VALU instructions do not execute
// next to an MFMA, whichever wave issues them (profiles/r04g_mfma_overlap_bf16.txt); the loop is 96 MFMAs + ~700 VALU in any order.
// (Round 5 corrected that rule -- DESIGN.md section 3.1e, profiles/r05a_mfma_fillers.txt: <= 5 plain VALU instructions per MFMA do hide when interleaved, packed
// f32 / v_dot2c do not, and this kernel's ~7.3 per MFMA with 190 packed ones is over either budget; its LDS round trips are the rest of the story.)
//
// Same op, same arithmetic and same derived filters as the SPLIT instantiation of conv3x3_wino.hip (plan option wino_split:
// every fp32 operand as x = h + m + l, six v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32 accumulation); what changes is
// who owns what.  The 8-wave kernel gives a wave two positions of the 4x4 transform domain and 256 registers, which leaves
// nothing in flight across its MFMA groups: its loop takes 9.9 k cycles per 16-channel chunk where the matrix pipe needs 3.1 k
// (profiles/r04c_wino_split_ablations.txt).  Here a workgroup is 4 waves, one per SIMD, each with the SIMD's whole register file:
//   * wave i owns transform ROW i -- the four positions (i, 0..3) x 64 tiles x 64 output channels = 256 accumulator registers
//     (AGPRs); the column pass of B^T d B gives all four positions from ONE row pass (8 LDS reads per half unit and four
//     positions, instead of 6 per two);
//   * the 256 architectural registers hold the chunk's 24 U fragments (refilled per position right after its last MFMA of the
//     chunk), the row-pass results of the current unit and the LDS reads / row pass of the NEXT unit, issued a whole unit ahead;
//   * the epilogue folds the four columns in registers (A^T = [1 1 1 0; 0 1 -1 -1]), so only the row combination goes through
//     LDS: 8 planes written and each read once (192 KB per tile instead of 640).
// One image per tile (16 x 16 output pixels), maps >= 16 x 16, no dropout form: everything else stays on conv3x3_wino.hip.
#include <stdlib.h>

#include <algorithm>

#include "sr3_common.h"

#ifndef SR3_EXPERIMENTS
// default build: the experiment is not compiled (csrc/build.sh -DSR3_EXPERIMENTS brings it back, with the opt-in halo split tiles)
namespace sr3 {
int conv3x3_wino4_forward(const ConvParams&, const WinoGeom&, const float*, hipStream_t) {
  set_error("conv: the four-wave split Winograd kernel (tile 13 / plan option wino4) is an experiment: build with -DSR3_EXPERIMENTS");
  return SR3_E_UNSUPPORTED;
}
}  // namespace sr3
#else
namespace sr3 {

typedef __bf16 qbf16x8 __attribute__((ext_vector_type(8)));
typedef int qint2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int QBN = 64;          // output channels per workgroup
constexpr int QCK = 16;          // input channels per chunk
constexpr int QRS = 20;          // LDS pixel stride (floats): 16 channels + 4 pad
constexpr int QTW = 18;          // raw halo pixels per row, and rows
constexpr int QROW = QTW * QRS + 8;
constexpr int QSHIFT = 4;        // every second row pair is shifted by 4 floats (conflict-free transform reads)
constexpr int QNT = 256;         // threads (4 waves)
constexpr int QHP = QTW * QTW;   // 324 raw halo pixels
constexpr int QHI = (QHP * 4 + QNT - 1) / QNT;       // 6 float4 staging items per thread
constexpr int QRAW_F = QTW * QROW + 8;                // floats per raw buffer (two of them)
constexpr int QETS = 68;         // epilogue exchange: floats per channel row of a plane (64 tiles + 4 pad)
constexpr int QEPL = 2240;       // floats per plane (32 channel rows + the shift of channels >= 16)
constexpr int QEXCH_F = 4 * 2 * QEPL;                 // 4 waves x 2 (q) planes
constexpr int QTAB_F = 2 * QHI * QNT;                 // parked staging items: [item][thread] of (hinfo, pixel)
constexpr int Q_MAX_CK = 64;
constexpr int QCST_F = 64 + Q_MAX_CK * 2 * QCK;
constexpr int QDUMMY_F = QNT * 4;                     // one 16-byte slot per thread: where a switched-off staging write lands
static_assert(2 * QRAW_F + QTAB_F + QDUMMY_F <= QEXCH_F, "raw tiles + item table + dummy slots inside the exchange block's footprint");
constexpr int Q_SMEM = (QEXCH_F + 2 * QCST_F) * 4;    // 71,680 + 16,896 bytes
constexpr int QUS = 3 * 64 * 8;  // bf16 elements of one (position, n block) fragment group (conv3x3_wino.hip: WUS)

__device__ __forceinline__ float silu_q(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
}
__device__ __forceinline__ void split3x8q(const f32x4& lo, const f32x4& hi, qbf16x8& h, qbf16x8& m, qbf16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    bf16x2 hh, mm, ll;
    split3_pair(e < 4 ? lo[e] : hi[e - 4], e < 4 ? lo[e + 1] : hi[e - 3], hh, mm, ll);       // (sr3_common.h)
    h[e] = hh[0]; h[e + 1] = hh[1]; m[e] = mm[0]; m[e + 1] = mm[1]; l[e] = ll[0]; l[e + 1] = ll[1];
  }
}
}  // namespace

// ACT: the prologue of the input (0 none, 1 GroupNorm affine, 2 affine + SiLU) as a template argument: the staging step must be
// one basic block to be interleaved with an MFMA group
template <int DBG, int ACT>
__global__ __launch_bounds__(QNT) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_conv3x3_wino4(const ConvParams p, const WinoGeom g, const __bf16* __restrict__ ufrag) {
  extern __shared__ f32x4 smem_q[];
  float* smem = reinterpret_cast<float*>(smem_q);
  float* raw0 = smem;
  float* raw1 = smem + QRAW_F;
  int* ptab = reinterpret_cast<int*>(smem + 2 * QRAW_F);
  float* cst = smem + QEXCH_F;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = transform row i
  const int Cin = p.C0 + p.C1;
  const int H = p.Ho, W = p.Wo;
  const int sp_tiles = g.tiles_w * g.tiles_h * g.nbt;
  const int ntiles = ((p.Cout + QBN - 1) / QBN) * sp_tiles;
  int cb = 0, tw_i = 0, th_i = 0, b0 = 0, h0 = 0, w0 = 0;
  auto decode_tile = [&](int v) {
    int bid = v;
    if ((ntiles & 7) == 0) bid = (bid & 7) * (ntiles >> 3) + (bid >> 3);      // one contiguous range of the list per XCD
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
    h0 = th_i * 16; w0 = tw_i * 16;
  };

  const int nch = (Cin + QCK - 1) / QCK;
  const int cper = (nch + p.ksplit - 1) / p.ksplit;
  const int c_begin = blockIdx.y * cper;
  const int c_end = min(nch, c_begin + cper);
  const int nck = c_end - c_begin;
  const bool direct = p.ksplit == 1;

  // ---- raw staging: item j of a thread covers halo pixel (tid >> 2) + 64 j, channel quad tid & 3 ----
  const int kq = tid & 3;
  int hinfo_r[QHI], hpix[QHI];      // only live between set_items / fetch_items and the staging step that follows
  auto set_items = [&]() {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int j = 0; j < QHI; ++j) {
      const int hp = (t_ >> 2) + (QNT / 4) * j;
      const int hy = hp / QTW, hx = hp - hy * QTW;
      hinfo_r[j] = hp < QHP ? (hy * QROW + ((hy >> 1) & 1) * QSHIFT + hx * QRS) : -1;
      const int ih = h0 + hy - 1, iw = w0 + hx - 1;
      const bool ok = hp < QHP && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      hpix[j] = ok ? (b0 * p.Hs + (ih >> p.ups)) * p.Ws + (iw >> p.ups) : -1;
    }
  };
  auto park_items = [&]() {
#pragma unroll
    for (int j = 0; j < QHI; ++j) reinterpret_cast<qint2*>(ptab)[j * QNT + tid] = qint2{hinfo_r[j], hpix[j]};
  };
  f32x4 rh[QHI];            // staging registers of the main loop (and of a tile's chunk 0)
  f32x4 rh2[QHI];           // ... of a tile's chunk 1: fetched during the previous tile's epilogue, idle in the main loop
  auto load_raw = [&](int chunk, f32x4 (&rh)[QHI]) {
    const int c = chunk * QCK + kq * 4;
    const int ce = c < Cin ? c : 0;
    const bool second = ce >= p.C0;
    const float* sp_ = second ? p.src1 : p.src0;
    const int sC = second ? p.C1 : p.C0;
    const int cs = second ? ce - p.C0 : ce;
#pragma unroll
    for (int j = 0; j < QHI; ++j) {
      const int off = hpix[j] >= 0 ? hpix[j] * sC + cs : 0;
      rh[j] = *reinterpret_cast<const f32x4*>(sp_ + off);
    }
  };
  auto store_raw = [&](float* raw, int chunk, const f32x4 (&rh)[QHI], const float* cs_) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const bool hvalid = chunk * QCK + kq * 4 < Cin;
    f32x4 ssa = zero, ssb = zero;
    if (ACT != 0) {
      const float* q = cs_ + 64 + (chunk - c_begin) * (2 * QCK) + kq * 8;
      ssa = *reinterpret_cast<const f32x4*>(q);
      ssb = *reinterpret_cast<const f32x4*>(q + 4);
    }
#pragma unroll
    for (int j = 0; j < QHI; ++j) {
      if (hinfo_r[j] >= 0) {
        f32x4 v = rh[j];
        if (ACT != 0) {
          v.x = fmaf(v.x, ssa.x, ssa.y);
          v.y = fmaf(v.y, ssa.z, ssa.w);
          v.z = fmaf(v.z, ssb.x, ssb.y);
          v.w = fmaf(v.w, ssb.z, ssb.w);
          if (ACT == 2) { v.x = silu_q(v.x); v.y = silu_q(v.y); v.z = silu_q(v.z); v.w = silu_q(v.w); }
        }
        v = (hvalid && hpix[j] >= 0) ? v : zero;
        *reinterpret_cast<f32x4*>(&raw[hinfo_r[j] + kq * 4]) = v;
      }
    }
  };
  // Main-loop form of the staging step, in two halves (items 3 h .. 3 h + 2) and WITHOUT branches, so that it can sit inside
  // an MFMA group's scheduling region and run in the gaps of the MFMA stream: `on` (wave-uniform: there is a chunk to stage)
  // only selects the LDS address -- a switched-off write lands in the thread's dummy slot -- and the chunk of the next global
  // load is clamped by the caller.  Items are re-read from their LDS table, the (scale, shift) pairs from the constants block.
  float* dummy_slot = smem + 2 * QRAW_F + QTAB_F + tid * 4;
  auto stage_half = [&](int h, float* raw, bool on, int chunk, int next_chunk, const float* cs_) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const bool hvalid = chunk * QCK + kq * 4 < Cin;
    f32x4 ssa = zero, ssb = zero;
    if (ACT != 0) {
      const float* q = cs_ + 64 + (chunk - c_begin) * (2 * QCK) + kq * 8;
      ssa = *reinterpret_cast<const f32x4*>(q);
      ssb = *reinterpret_cast<const f32x4*>(q + 4);
    }
    const int cnx = next_chunk * QCK + kq * 4;
    const int ce = cnx < Cin ? cnx : 0;
    const bool second = ce >= p.C0;
    const float* sp_ = second ? p.src1 : p.src0;
    const int sC = second ? p.C1 : p.C0;
    const int cs = second ? ce - p.C0 : ce;
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int j = 3 * h + jj;
      const qint2 it = reinterpret_cast<const qint2*>(ptab)[j * QNT + tid];      // (LDS offset | -1, source pixel | -1)
      f32x4 v = rh[j];
      if (ACT != 0) {
        v.x = fmaf(v.x, ssa.x, ssa.y);
        v.y = fmaf(v.y, ssa.z, ssa.w);
        v.z = fmaf(v.z, ssb.x, ssb.y);
        v.w = fmaf(v.w, ssb.z, ssb.w);
        if (ACT == 2) { v.x = silu_q(v.x); v.y = silu_q(v.y); v.z = silu_q(v.z); v.w = silu_q(v.w); }
      }
      v = (hvalid && it.y >= 0) ? v : zero;
      float* dstp = (on && it.x >= 0) ? raw + it.x + kq * 4 : dummy_slot;
      *reinterpret_cast<f32x4*>(dstp) = v;
      const int off = it.y >= 0 ? it.y * sC + cs : 0;
      rh[j] = *reinterpret_cast<const f32x4*>(sp_ + off);
    }
  };
  // per-tile constants: bias + FiLM row of the 64 output channels (cst[0..63]), (scale, shift) pairs of the split's channels
  const float* dummy = p.w;
  auto stage_consts = [&](float* cs_) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (tid < 16) {
      const int n = cb * QBN + tid * 4;
      f32x4 v = zero;
      if (direct && n < p.Cout) {
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.film) v += *reinterpret_cast<const f32x4*>(p.film + (size_t)b0 * p.film_stride + n);
      }
      *reinterpret_cast<f32x4*>(cs_ + tid * 4) = v;
    }
    const float* q = ACT != 0 ? p.ss + (size_t)b0 * Cin * 2 : dummy;
    for (int e = tid; e < nck * (2 * QCK) / 4; e += QNT) {
      const int ch = c_begin * QCK + e * 2;
      f32x4 v = zero;
      if (ACT != 0 && ch < Cin) v = *reinterpret_cast<const f32x4*>(q + (size_t)ch * 2);
      *reinterpret_cast<f32x4*>(cs_ + 64 + e * 4) = v;
    }
  };

  // ---- this wave's transform row: t = d[ra] + sgn * d[rb] ----
  const int wi = wave;
  const int ra = (wi == 0) ? 0 : (wi == 2 ? 2 : 1);
  const int rb = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
  const float rsgn = (wi == 1) ? 1.f : -1.f;
  const int tl = lane & 31, hq = lane >> 5;
  const int tyl = tl >> 3, tx = tl & 7;
  const int r0 = 2 * tyl;
  const int offa = (r0 + ra) * QROW + (((r0 + ra) >> 1) & 1) * QSHIFT + (2 * tx) * QRS + hq * 4;
  const int offb = (r0 + rb) * QROW + (((r0 + rb) >> 1) & 1) * QSHIFT + (2 * tx) * QRS + hq * 4;
  // row pass of one half unit (tile block m, half chunk kk): eight reads, four patch columns
  auto rowpass = [&](const float* rawbuf, int m, int kk, f32x4 (&t)[4]) {
    if (DBG & 4) {                // ablation: no LDS reads, no row pass
#pragma unroll
      for (int s = 0; s < 4; ++s) t[s] = f32x4{1.f, 2.f, 3.f, 4.f};
      return;
    }
    f32x4 da[4], db[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      da[s] = *reinterpret_cast<const f32x4*>(rawbuf + offa + (m * 8 * QROW + s * QRS + kk * 8));
      db[s] = *reinterpret_cast<const f32x4*>(rawbuf + offb + (m * 8 * QROW + s * QRS + kk * 8));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) t[s] = da[s] + db[s] * rsgn;
  };
  // column pass: position (i, j) from the row-pass results of one half chunk
  auto colpass = [&](const f32x4 (&t)[4], int j) -> f32x4 {
    return j == 0 ? t[0] - t[2] : (j == 1 ? t[1] + t[2] : (j == 2 ? t[2] - t[1] : t[1] - t[3]));
  };

  // ---- U fragments: [pj][nblk][plane], one whole chunk; refilled per position after its last MFMA of the chunk ----
  qbf16x8 us[4][2][3];
  const __bf16* ubase = nullptr;
  auto load_us = [&](int chunk, int pj) {
    if ((DBG & 16) && chunk != c_begin) return;
    const __bf16* q = ubase + (size_t)chunk * 16 * (2 * QUS) + pj * (2 * QUS);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) us[pj][n][pl] = *reinterpret_cast<const qbf16x8*>(q + n * QUS + pl * 512);
  };
  f32x16 acc[4][2][2];          // [pj][mblk][nblk]
  // One position of a unit: its 12 MFMAs (operand planes `vc`, built a position earlier) with the 3 x bf16 split of the NEXT
  // position's operands (-> `vn`) interleaved between them: one wave per SIMD, so the VALU work has to sit inside the MFMA stream
  // in program order to run beside it (sched_group_barrier: one MFMA, then up to four VALU instructions, twelve times).
  qbf16x8 vc[3], vn[3];
  auto mfma_pos = [&](int m, int pj, bool has_next, const f32x4& nlo, const f32x4& nhi, auto&& filler, int valu_per_mfma) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // smallest terms first
    __builtin_amdgcn_sched_barrier(0);
    filler();                     // independent work that shares this group's scheduling region (the staging halves)
    if (has_next) {
      if (DBG & 128) {            // ablation: one conversion per value instead of the 3-way split
#pragma unroll
        for (int e = 0; e < 8; ++e) vn[0][e] = (__bf16)(e < 4 ? nlo[e] : nhi[e - 4]);
        vn[1] = vn[0]; vn[2] = vn[0];
      } else {
        split3x8q(nlo, nhi, vn[0], vn[1], vn[2]);
      }
    }
    if (DBG & 1) {                // ablation: no MFMAs (operands kept live)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) acc[pj][m][n][pl] += (float)vc[pl][0] * (float)us[pj][n][pl][0];
    } else {
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[pj][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc[PA[q]], us[pj][n][PB[q]], acc[pj][m][n], 0, 0, 0);
    }
#ifndef SR3_W4_NO_INTERLEAVE
    if (valu_per_mfma <= 4) {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // up to four VALU
      }
    } else {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);      // up to seven VALU ...
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);      // ... and two transcendentals (the staging half's SiLU)
      }
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) { vc[0] = vn[0]; vc[1] = vn[1]; vc[2] = vn[2]; }
  };

  const int T = g.tiles_h * g.tiles_w;
  const bool stats = direct && p.ostat != nullptr;
  const bool has_res = direct && p.res0 != nullptr;
  const size_t Mtot = (size_t)p.B * H * W;
  float* dst = direct ? p.out : p.partial + (size_t)blockIdx.y * Mtot * p.Cout;

  // Persistent workgroups: what a tile needs before its main loop -- constants, staging items, raw chunks 0 and 1 -- is fetched
  // during the previous tile's epilogue (the last tile re-fetches itself: the loads stay unconditional).
  int vtile = blockIdx.x;
  int par = 0;
  const int c1 = min(c_begin + 1, c_end - 1);
  decode_tile(vtile);
  set_items();
  load_raw(c_begin, rh);
  load_raw(c1, rh2);
  stage_consts(cst);
  for (;;) {
    // ================================ prologue ================================
    const float* cs_ = cst + par * QCST_F;
    ubase = ufrag + (size_t)cb * nch * 16 * (2 * QUS) + (size_t)(wi * 4) * (2 * QUS) + lane * 8;
#pragma unroll
    for (int pj = 0; pj < 4; ++pj) load_us(c_begin, pj);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;
    __syncthreads();                       // the constants are visible, the previous tile's epilogue is done with LDS
    park_items();
    store_raw(raw0, c_begin, rh, cs_);
    if (nck > 1) store_raw(raw1, c_begin + 1, rh2, cs_);
    if (nck > 2) load_raw(c_begin + 2, rh);
    __syncthreads();

    // ================================ main loop ================================
    // Units are tile blocks m of a chunk.  tt[kk][s]: row-pass results of the CURRENT unit; tn: of the next one, built during
    // the current unit's MFMAs (first half chunk behind position 1, second behind position 3).
    f32x4 tt[2][4], tn[2][4];
    rowpass(raw0, 0, 0, tt[0]);
    rowpass(raw0, 0, 1, tt[1]);
    split3x8q(colpass(tt[0], 0), colpass(tt[1], 0), vc[0], vc[1], vc[2]);
    for (int i = 0; i < nck; ++i) {
      float* rcur = (i & 1) ? raw1 : raw0;
      const float* rnext = (i & 1) ? raw0 : raw1;
      const bool more = i + 1 < nck;
      // ---- unit (i, m0); next: (i, m1) from rcur ----
      auto none = [] {};
      mfma_pos(0, 0, true, colpass(tt[0], 1), colpass(tt[1], 1), none, 4);
      mfma_pos(0, 1, true, colpass(tt[0], 2), colpass(tt[1], 2), none, 4);
      rowpass(rcur, 1, 0, tn[0]);
      mfma_pos(0, 2, true, colpass(tt[0], 3), colpass(tt[1], 3), none, 4);
      rowpass(rcur, 1, 1, tn[1]);
      mfma_pos(0, 3, true, colpass(tn[0], 0), colpass(tn[1], 0), none, 4);      // (next position = position 0 of unit m1)
#pragma unroll
      for (int s = 0; s < 4; ++s) { tt[0][s] = tn[0][s]; tt[1][s] = tn[1][s]; }
      __syncthreads();                     // rcur is fully consumed: chunk i + 2 goes into it; chunk i + 1 is visible in rnext
      // ---- unit (i, m1); next: (i + 1, m0) from rnext; U of chunk i + 1 per position; chunk i + 2 is staged into rcur in the
      // gaps of the first two MFMA groups.  Straight-line code (the interleaving directives work on one basic block): the last
      // chunks re-fetch their own U fragments / raw data, transform a stale raw tile and stage into a dummy slot, and nothing reads
      // the results ----
      const int cn = c_begin + (more ? i + 1 : i);
      const bool on = i + 2 < nck && !(DBG & 32);
      const int cst_chunk = c_begin + min(i + 2, nck - 1), nxt_chunk = c_begin + min(i + 3, nck - 1);
      mfma_pos(1, 0, true, colpass(tt[0], 1), colpass(tt[1], 1), [&] { if (!(DBG & 32)) stage_half(0, rcur, on, cst_chunk, nxt_chunk, cs_); }, 9);
      load_us(cn, 0);
      mfma_pos(1, 1, true, colpass(tt[0], 2), colpass(tt[1], 2), [&] { if (!(DBG & 32)) stage_half(1, rcur, on, cst_chunk, nxt_chunk, cs_); }, 9);
      load_us(cn, 1);
      rowpass(rnext, 0, 0, tn[0]);
      mfma_pos(1, 2, true, colpass(tt[0], 3), colpass(tt[1], 3), none, 4);
      load_us(cn, 2);
      rowpass(rnext, 0, 1, tn[1]);
      mfma_pos(1, 3, true, colpass(tn[0], 0), colpass(tn[1], 0), none, 4);
      load_us(cn, 3);
#pragma unroll
      for (int s = 0; s < 4; ++s) { tt[0][s] = tn[0][s]; tt[1][s] = tn[1][s]; }
    }

    // ================================ epilogue ================================
    // fold the four columns with A (A^T = [1 1 1 0; 0 1 -1 -1]):  P_0 = M0 + M1 + M2,  P_1 = M1 - M2 - M3  (registers), then
    // Y[p][q] = sum_i A^T[p][i] P_q(i) through LDS in a FIXED order: p = 0: (r0 + r1) + r2, p = 1: (r1 - r2) - r3.  Plane (wave i,
    // q) = [32 channels][64 tiles + 4 pad], channels >= 16 shifted by 4 floats (the layout of conv3x3_wino.hip); one 32-channel
    // block per round; thread -> (q, 4 consecutive tiles, 4 consecutive channels), both p.
    const int e_cb = cb, e_b0 = b0, e_tix = th_i * g.tiles_w + tw_i, e_vtile = vtile;
    const int nq = lane & 7, fq = (lane >> 4) & 1;
    const int tq = ((lane >> 5) & 1) | (((lane >> 3) & 1) << 1) | (wave << 2);      // tiles 4 tq .. 4 tq + 3 (half a tile row)
    const int ety = tq >> 1, etx0 = (tq & 1) * 4;
    const size_t pixp0 = ((size_t)b0 * H + (h0 + 2 * ety)) * W + (w0 + 2 * etx0 + fq);      // p = 0; p = 1: + W; tile k: + 2 k
    f32x4 base[2];
#pragma unroll
    for (int nblk = 0; nblk < 2; ++nblk) base[nblk] = *reinterpret_cast<const f32x4*>(cs_ + nblk * 32 + nq * 4);
    // this tile's residual, both rounds and both p: 16 loads put in flight before the exchange (absent: a valid dummy address)
    f32x4 addv[2][2][4];
#pragma unroll
    for (int nblk = 0; nblk < 2; ++nblk) {
      const int n = e_cb * QBN + nblk * 32 + nq * 4;
      const int ne = n < p.Cout ? n : 0;
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const size_t pix = pixp0 + (size_t)pp * W + 2 * k;
          const float* rp = dst + pix * p.Cout + ne;
          if (has_res) rp = (ne < p.RC0) ? p.res0 + pix * p.RC0 + ne : p.res1 + pix * p.RC1 + (ne - p.RC0);
          addv[nblk][pp][k] = *reinterpret_cast<const f32x4*>(rp);
        }
    }
    if (DBG & 8) {                                     // ablation: no epilogue (one store keeps the accumulators live)
      float sacc = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += acc[a][b][c][r];
      p.out[(size_t)e_vtile * QNT + tid] = sacc + addv[0][0][0][0];
      const bool nxt = vtile + (int)gridDim.x < ntiles;
      if (nxt) vtile += gridDim.x;
      __syncthreads();
      decode_tile(vtile);
      set_items();
      load_raw(c_begin, rh);
      load_raw(c1, rh2);
      stage_consts(cst + (par ^ 1) * QCST_F);
      par ^= 1;
      if (!nxt) break;
      continue;
    }
    const bool has_next = vtile + (int)gridDim.x < ntiles;
    if (has_next) vtile += gridDim.x;
    __syncthreads();                                   // the raw tiles and the item table are dead: the exchange block reuses LDS
    float* exch = smem;
    double s1[2][4], s2[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int k = 0; k < 4; ++k) { s1[a][k] = 0.0; s2[a][k] = 0.0; }
    const int wn = lane & 31;
    float* wbase = exch + (wave * 2) * QEPL + wn * QETS + (wn >> 4) * 4 + 4 * (lane >> 5);
#pragma unroll
    for (int nblk = 0; nblk < 2; ++nblk) {
      const int n = e_cb * QBN + nblk * 32 + nq * 4;
      const bool nok = n < p.Cout;
#pragma unroll
      for (int mblk = 0; mblk < 2; ++mblk) {
        const f32x16 p0 = (acc[0][mblk][nblk] + acc[1][mblk][nblk]) + acc[2][mblk][nblk];
        const f32x16 p1 = (acc[1][mblk][nblk] - acc[2][mblk][nblk]) - acc[3][mblk][nblk];
        // D layout: reg r of lane l -> tile (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the block, channel l & 31
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          *reinterpret_cast<f32x4*>(wbase + mblk * 32 + 8 * k) = f32x4{p0[4 * k], p0[4 * k + 1], p0[4 * k + 2], p0[4 * k + 3]};
          *reinterpret_cast<f32x4*>(wbase + QEPL + mblk * 32 + 8 * k) = f32x4{p1[4 * k], p1[4 * k + 1], p1[4 * k + 2], p1[4 * k + 3]};
        }
      }
      if (nblk == 0) {
        // the next tile's constants, staging items and raw chunks 0 and 1 are put in flight (half of the accumulators are dead)
        decode_tile(vtile);
        set_items();
        load_raw(c_begin, rh);
        load_raw(c1, rh2);
        stage_consts(cst + (par ^ 1) * QCST_F);
      }
      __syncthreads();
      {
        f32x4 y[2][4];                                               // [p][channel j] over the 4 tiles
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nn = nq * 4 + j;
          const float* rb_ = exch + nn * QETS + (nn >> 4) * 4 + 4 * tq + fq * QEPL;
          const f32x4 r0_ = *reinterpret_cast<const f32x4*>(rb_ + 0 * 2 * QEPL);
          const f32x4 r1_ = *reinterpret_cast<const f32x4*>(rb_ + 1 * 2 * QEPL);
          const f32x4 r2_ = *reinterpret_cast<const f32x4*>(rb_ + 2 * 2 * QEPL);
          const f32x4 r3_ = *reinterpret_cast<const f32x4*>(rb_ + 3 * 2 * QEPL);
          y[0][j] = (r0_ + r1_) + r2_;
          y[1][j] = (r1_ - r2_) - r3_;
        }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const size_t pix = pixp0 + (size_t)pp * W + 2 * k;
            f32x4 v = f32x4{y[pp][0][k], y[pp][1][k], y[pp][2][k], y[pp][3][k]};
            if (direct) {
              v += base[nblk];
              if (has_res) v += addv[nblk][pp][k];
              if (stats) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { const double dv = (double)v[c]; s1[nblk][c] += dv; s2[nblk][c] += dv * dv; }
              }
            }
            if (nok) *reinterpret_cast<f32x4*>(dst + pix * p.Cout + n) = v;
          }
      }
      __syncthreads();                                  // every read of the exchange block is complete
    }
    if (stats) {
      // per-channel sums of this tile's outputs in a fixed order: part[e][thread] (e = [sum | sumsq][32-channel block][channel of
      // the quad]), 256 threads each add 16 of the 32 partials that share a channel quad (threads nq, nq + 8, ...), 128 add two
      double* part = reinterpret_cast<double*>(smem);
      constexpr int PR = 264;
#pragma unroll
      for (int nb2 = 0; nb2 < 2; ++nb2)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          part[(nb2 * 4 + k) * PR + tid] = s1[nb2][k];
          part[(8 + nb2 * 4 + k) * PR + tid] = s2[nb2][k];
        }
      __syncthreads();
      {
        const int cq = tid & 7, e = (tid >> 3) & 15, grp = tid >> 7;
        double a = 0.0;
#pragma unroll
        for (int s = 0; s < 16; ++s) a += part[e * PR + (grp * 16 + s) * 8 + cq];
        part[16 * PR + grp * 128 + e * 8 + cq] = a;
      }
      __syncthreads();
      if (tid < 128) {
        const int cq = tid & 7, e = tid >> 3;             // e = which * 8 + nb2 * 4 + k
        const double a = part[16 * PR + tid] + part[16 * PR + 128 + tid];
        const int which = e >> 3, c = ((e >> 2) & 1) * 32 + cq * 4 + (e & 3);
        const int nn = e_cb * QBN + c;
        if (nn < p.Cout) p.ostat[(((size_t)e_b0 * T + e_tix) * p.Cout + nn) * 2 + which] = a;
      }
    }
    par ^= 1;
    if (!has_next) break;
  }
}

// ---- host -----------------------------------------------------------------------------------------------------
int conv3x3_wino4_forward(const ConvParams& p, const WinoGeom& g, const float* ufrag, hipStream_t st) {
  if (g.NB != 1 || p.drop_thresh != 0) { set_error("conv: the four-wave Winograd kernel covers the one-image tile without dropout"); return SR3_E_UNSUPPORTED; }
  static const int n_cu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n & ~7;
  }();
  const long ntiles = wino_workgroups(p, g);
  dim3 grid((unsigned)std::min<long>(ntiles, n_cu > 0 ? n_cu : 256), p.ksplit);
  static const int dbg = [] { const char* e = getenv("SR3_WINO_DBG"); return e ? atoi(e) : 0; }();
#define SR3_W4_LAUNCH2(D, A)                                                                                              \
  {                                                                                                                   \
    static std::atomic<uint64_t> done{0};                                                                             \
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(k_conv3x3_wino4<D, A>), Q_SMEM, done)) return rc;       \
    hipLaunchKernelGGL((k_conv3x3_wino4<D, A>), grid, dim3(QNT), Q_SMEM, st, p, g, reinterpret_cast<const __bf16*>(ufrag)); \
  }
#define SR3_W4_LAUNCH(D) { if (p.act == 2) SR3_W4_LAUNCH2(D, 2) else if (p.act == 1) SR3_W4_LAUNCH2(0, 1) else SR3_W4_LAUNCH2(0, 0) }   /* (ablations: act = 2 layers) */
  switch (dbg) {
    case 0:
      if (p.act == 2) SR3_W4_LAUNCH2(0, 2) else if (p.act == 1) SR3_W4_LAUNCH2(0, 1) else SR3_W4_LAUNCH2(0, 0)
      break;
#ifdef SR3_WINO_ABLATIONS
    case 1: SR3_W4_LAUNCH(1) break;
    case 4: SR3_W4_LAUNCH(4) break;
    case 8: SR3_W4_LAUNCH(8) break;
    case 16: SR3_W4_LAUNCH(16) break;
    case 32: SR3_W4_LAUNCH(32) break;
    case 128: SR3_W4_LAUNCH(128) break;
    case 180: SR3_W4_LAUNCH(180) break;        // 4 + 16 + 32 + 128: the bare MFMA loop + prologue / epilogue
    case 181: SR3_W4_LAUNCH(181) break;        // ... without the MFMAs: prologue / epilogue only
#endif
    default: set_error("conv: SR3_WINO_DBG=%d is not built for the four-wave kernel", dbg); return SR3_E_BADARG;
  }
#undef SR3_W4_LAUNCH
#undef SR3_W4_LAUNCH2
  SR3_LAUNCH_CHECK("k_conv3x3_wino4");
  return SR3_OK;
}

}  // namespace sr3

#endif  // SR3_EXPERIMENTS
