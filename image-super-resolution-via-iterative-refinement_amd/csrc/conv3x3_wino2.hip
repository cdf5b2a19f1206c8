// Winograd F(2x2, 3x3) on the bf16 MFMA with 3-way split operands: TWO independent four-wave workgroups per CU (round 6).
//
// Same op, same arithmetic and same derived filters as the SPLIT instantiation of conv3x3_wino.hip (`Block` = GroupNorm -> Swish ->
// Conv3x3 of model/sr3_modules/unet.py:80-91, FiLM add :34-50,108, residual add :110, Upsample's conv + nearest x2 :58-65, the skip
// concat :255; every fp32 operand as x = h + m + l, six v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32 accumulation).  What
// changes is who owns what, for two measured reasons:
//   * the 8-wave kernel is ONE workgroup per CU whose waves run in lock step (one barrier per chunk): while it is in a tile's
//     epilogue / prologue (17.5 k + 3.5 k cycles per tile, 27 % of a K = 64 tile at 128 x 128) the matrix pipe of the CU is idle, and the
//     two waves of a SIMD sit in their MFMA groups together and in their VALU / LDS sections together (MFMA busy 0.28).  Here a
//     workgroup is 4 waves x 256 registers and <= 80 KB of LDS, so a CU holds TWO of them that know nothing of each other: one's
//     epilogue, barrier waits and LDS round trips run under the other's MFMA groups, and the two waves of a SIMD belong to
//     different workgroups and drift out of phase by themselves;
//   * a wave owns one transform COLUMN j and all four rows i: 4 positions x 32 tiles x 64 output channels = 128 accumulators.  The
//     row half of A^T M A (4 -> 2) is then folded in registers and only the column combination crosses LDS: 8 b128 writes + 12 reads
//     per thread and 32-channel round instead of 16 + 24, and the column pass of B^T d B is shared by the four positions: 16 LDS reads
//     per chunk and wave instead of 24.
// Workgroup tile: 32 Winograd tiles = 8 x 16 output pixels of one image (raw halo 10 x 18) x 64 output channels x all 16 positions.
// The price is the filter traffic from L2 (a U fragment serves 32 tiles instead of 64); U never goes through LDS, as before.
// One-image tile, maps >= 16 wide and a multiple of 8 high, no dropout form: everything else stays on conv3x3_wino.hip.
#include <stdlib.h>

#include <algorithm>

#include "sr3_common.h"

namespace sr3 {

typedef int vint2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int VBN = 64;          // output channels per workgroup
constexpr int VCK = 16;          // input channels per chunk
constexpr int VRS = 20;          // LDS pixel stride (floats): 16 channels + 4 pad
constexpr int VTW = 18;          // raw halo pixels per row
constexpr int VTH = 10;          // raw halo rows
constexpr int VROW = VTW * VRS + 8;
constexpr int VSHIFT = 4;        // every second row pair is shifted by 4 floats (conflict-free transform reads, as in the 8-wave kernel)
constexpr int VNT = 256;         // threads (4 waves)
constexpr int VHP = VTH * VTW;   // 180 raw halo pixels
constexpr int VHI = (VHP * 4 + VNT - 1) / VNT;       // 3 float4 staging items per thread
constexpr int VRAW_F = VTH * VROW + 8;                // floats per raw buffer (two of them)
constexpr int VTAB_F = 2 * VHI * VNT;                 // parked staging items: [item][thread] of (hinfo, pixel)
constexpr int VWORK_F = 9504;                         // the main loop's LDS (raw tiles + item table), the exchange block and the statistics tree share it
constexpr int VETS = 36;         // epilogue exchange: floats per channel row of a plane (32 tiles + 4 pad)
constexpr int VEPL = 32 * VETS + 4;                   // floats per plane (32 channel rows; channels >= 16 shifted by 4 floats)
constexpr int VEXCH_F = 4 * 2 * VEPL;                 // 4 waves x 2 (p) planes of one 32-channel round
constexpr int VPR = 264;                              // statistics parking: doubles per row (256 threads + 8)
constexpr int V_MAX_CK = 64;                          // chunks of one workgroup's K range (1024 input channels; more: split-K)
constexpr int VCST_F = 64 + V_MAX_CK * 2 * VCK;       // per-tile constants (bias + FiLM of 64 channels, (scale, shift) pairs), two of them
static_assert(2 * VRAW_F + VTAB_F <= VWORK_F && VEXCH_F <= VWORK_F && (16 * VPR + 256) * 2 <= VWORK_F, "raw tiles + item table, exchange block, statistics tree");
constexpr int V_SMEM = (VWORK_F + 2 * VCST_F) * 4;    // 54,912 bytes: two workgroups per CU
static_assert(V_SMEM <= 81920, "LDS: two workgroups per CU");
constexpr int VUS = 3 * 64 * 8;  // bf16 elements of one (position, n block) fragment group (conv3x3_wino.hip: WUS)

__device__ __forceinline__ float silu_v(float v) {      // (conv3x3_wino.hip: silu_w)
#ifdef SR3_EXACT_ACT
  return SR3_SILU(v);
#else
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
#endif
}
}  // namespace

// DBG (profiling only, env SR3_WINO_DBG, library built with -DSR3_WINO_ABLATIONS; 0 in production): 1 skip the MFMAs, 4 skip the input
// transform, 8 skip the epilogue, 16 skip the U loads of the loop, 32 skip the raw staging of the loop, 128 one conversion instead of
// the 3-way split.
// Global loads outside the main loop are unconditional and consumed in issue order (absent operands: a valid dummy address,
// discarded by a select) -- the in-order vmcnt rule of conv3x3_wino.hip.
template <int DBG>
__global__ __launch_bounds__(VNT, 2) void k_conv3x3_wino2(const ConvParams p, const WinoGeom g, const __bf16* __restrict__ ufrag) {
  extern __shared__ f32x4 smem_w2[];
  float* smem = reinterpret_cast<float*>(smem_w2);
  float* raw0 = smem;
  float* raw1 = smem + VRAW_F;
  int* ptab = reinterpret_cast<int*>(smem + 2 * VRAW_F);
  float* cst = smem + VWORK_F;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = transform column j (an SGPR: what depends on it is scalar)
  // lane / thread id re-derived where they are needed (two VALU instructions, volatile: neither hoisted out of the tile loop nor kept --
  // or spilled: a scratch reload waits on vmcnt(0), i.e. on every U fragment in flight -- in a register across the MFMA groups)
  auto lane_now = [&]() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  auto tid_now = [&]() { return wave * 64 + lane_now(); };
  const int Cin = p.C0 + p.C1;
  const int H = p.Ho, W = p.Wo;
  // persistent workgroups, one contiguous range of the (cout block major) tile list per XCD: conv3x3_wino.hip
  const int sp_tiles = g.tiles_w * g.tiles_h * g.nbt;
  const int ntiles = ((p.Cout + VBN - 1) / VBN) * sp_tiles;
  int cb = 0, tw_i = 0, th_i = 0, b0 = 0, h0 = 0, w0 = 0;
  auto decode_tile = [&](int v) {
    int bid = v;
    if ((ntiles & 7) == 0) bid = (bid & 7) * (ntiles >> 3) + (bid >> 3);
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
    h0 = th_i * 8; w0 = tw_i * 16;
  };

  const int nch = (Cin + VCK - 1) / VCK;
  const int cper = (nch + p.ksplit - 1) / p.ksplit;
  const int c_begin = blockIdx.y * cper;
  const int c_end = min(nch, c_begin + cper);
  const int nck = c_end - c_begin;                // 1 .. V_MAX_CK (host)
  const bool direct = p.ksplit == 1;

  // ---- raw staging: item j of a thread covers halo pixel (tid >> 2) + 64 j, channel quad tid & 3 ----
  // hinfo: LDS float offset (bits 0..15), -1: no item; hpix: source pixel (-1: zero padding).  Not live across the main loop: parked in
  // LDS ([item][thread], every thread reads only what it wrote) and re-read right before every staging step.
  int hinfo_r[VHI], hpix[VHI];
  auto set_items = [&]() {
    const int t_ = tid_now();
#pragma unroll
    for (int j = 0; j < VHI; ++j) {
      const int hp = (t_ >> 2) + (VNT / 4) * j;
      const int hy = hp / VTW, hx = hp - hy * VTW;
      hinfo_r[j] = hp < VHP ? (hy * VROW + ((hy >> 1) & 1) * VSHIFT + hx * VRS) : -1;
      const int ih = h0 + hy - 1, iw = w0 + hx - 1;
      const bool ok = hp < VHP && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      hpix[j] = ok ? (b0 * p.Hs + (ih >> p.ups)) * p.Ws + (iw >> p.ups) : -1;
    }
  };
  auto park_items = [&]() {
    const int t_ = tid_now();
#pragma unroll
    for (int j = 0; j < VHI; ++j) reinterpret_cast<vint2*>(ptab)[j * VNT + t_] = vint2{hinfo_r[j], hpix[j]};
  };
  auto fetch_items = [&]() {
    const int t_ = tid_now();
#pragma unroll
    for (int j = 0; j < VHI; ++j) {
      const vint2 v = reinterpret_cast<const vint2*>(ptab)[j * VNT + t_];
      hinfo_r[j] = v.x; hpix[j] = v.y;
    }
  };
  f32x4 rh[VHI];            // staging registers of the main loop (and of the tile's chunk 0)
  f32x4 rh2[VHI];           // ... of the tile's chunk 1: fetched during the previous tile's epilogue, idle in the main loop
  auto load_raw = [&](int chunk, f32x4 (&r)[VHI]) {
    const int t_ = tid_now();
    const int kq = t_ & 3;
    const int c = chunk * VCK + kq * 4;
    const int ce = c < Cin ? c : 0;
    const bool second = ce >= p.C0;
    const float* sp_ = second ? p.src1 : p.src0;
    const int sC = second ? p.C1 : p.C0;
    const int cs = second ? ce - p.C0 : ce;
#pragma unroll
    for (int j = 0; j < VHI; ++j) {
      const int hp_ = hpix[j];
      const int off = hp_ >= 0 ? hp_ * sC + cs : 0;
      r[j] = *reinterpret_cast<const f32x4*>(sp_ + off);
    }
  };
  auto store_raw = [&](float* raw, int chunk, const f32x4 (&r)[VHI], const float* cs_) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int t_ = tid_now();
    const int kq = t_ & 3;
    const bool hvalid = chunk * VCK + kq * 4 < Cin;
    f32x4 ssa = zero, ssb = zero;
    if (p.act != 0) {
      const float* q = cs_ + 64 + (chunk - c_begin) * (2 * VCK) + kq * 8;
      ssa = *reinterpret_cast<const f32x4*>(q);
      ssb = *reinterpret_cast<const f32x4*>(q + 4);
    }
#pragma unroll
    for (int j = 0; j < VHI; ++j) {
      if (hinfo_r[j] >= 0) {
        f32x4 v = r[j];
        if (p.act != 0) {
          v.x = fmaf(v.x, ssa.x, ssa.y);
          v.y = fmaf(v.y, ssa.z, ssa.w);
          v.z = fmaf(v.z, ssb.x, ssb.y);
          v.w = fmaf(v.w, ssb.z, ssb.w);
          if (p.act == 2) { v.x = silu_v(v.x); v.y = silu_v(v.y); v.z = silu_v(v.z); v.w = silu_v(v.w); }
        }
        v = (hvalid && hpix[j] >= 0) ? v : zero;
        *reinterpret_cast<f32x4*>(&raw[hinfo_r[j] + kq * 4]) = v;
      }
    }
  };
  // per-tile constants: bias + FiLM row of the tile's 64 output channels (cst[0..63]) and the (scale, shift) pairs of channels
  // [16 c_begin, 16 c_end) (cst[64..]).  Two unconditional loads + two of pairs per thread (nck <= 64: host), then LDS.
  const float* dummy = p.w;
  f32x4 creg[4];
  auto load_consts = [&]() {
    const int t_ = tid_now();
    const int n = cb * VBN + (t_ & 15) * 4;
    const int ne = n < p.Cout ? n : 0;
    creg[0] = *reinterpret_cast<const f32x4*>((direct && p.bias ? p.bias : dummy) + ne);
    creg[1] = *reinterpret_cast<const f32x4*>(direct && p.film ? p.film + (size_t)b0 * p.film_stride + ne : dummy);
    const float* q = p.act != 0 ? p.ss + (size_t)b0 * Cin * 2 : dummy;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = t_ + VNT * k;                                // 4 floats = 2 channels
      const int ch = c_begin * VCK + e * 2;
      creg[2 + k] = *reinterpret_cast<const f32x4*>(q + (p.act != 0 && ch < Cin ? (size_t)ch * 2 : 0));
    }
  };
  auto store_consts = [&](float* cs_) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int t_ = tid_now();
    if (t_ < 16) {
      const int n = cb * VBN + t_ * 4;
      f32x4 v = zero;
      if (direct && n < p.Cout) v = (p.bias ? creg[0] : zero) + (p.film ? creg[1] : zero);
      *reinterpret_cast<f32x4*>(cs_ + t_ * 4) = v;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = t_ + VNT * k;
      const int n8 = nck * (2 * VCK) / 4;
      const int ch = c_begin * VCK + e * 2;
      if (e < n8) *reinterpret_cast<f32x4*>(cs_ + 64 + e * 4) = (p.act != 0 && ch < Cin) ? creg[2 + k] : zero;
    }
  };

  // ---- this wave's transform column ----
  // B^T d B, column j:  c_r = d[r][ca] + sb * d[r][cb]  (j = 0: d0 - d2, 1: d1 + d2, 2: d1 - d2 = MINUS the true column, 3: d1 - d3), then
  // rows V_0 = c0 - c2, V_1 = c1 + c2, V_2 = c2 - c1, V_3 = c1 - c3.  Wave 2's sign flip costs nothing: the epilogue's column
  // combination subtracts its plane where the formula adds it and vice versa.
  // Transform lanes: lane l works on tile l & 31 of the 4 x 8 tile block and on channel quad l >> 5 of either half chunk -- the A
  // operand layout of the MFMA (row l & 31, k-half l >> 5): V never goes through LDS as an operand.
  const int ca = (wave == 0) ? 0 : 1, cb_ = (wave == 3) ? 3 : 2;
  const float sb = (wave == 1) ? 1.f : -1.f;
  const int tl = lane & 31, hq = lane >> 5;
  const int tyl = tl >> 3, tx = tl & 7;
  const int base01 = (2 * tyl) * VROW + (tyl & 1) * VSHIFT + 2 * tx * VRS + hq * 4;             // patch rows 0, 1
  const int base23 = (2 * tyl + 2) * VROW + ((tyl + 1) & 1) * VSHIFT + 2 * tx * VRS + hq * 4;   // patch rows 2, 3
  const int offa01 = base01 + ca * VRS, offb01 = base01 + cb_ * VRS, offa23 = base23 + ca * VRS, offb23 = base23 + cb_ * VRS;
  // One position (row i of this wave's column) and half chunk kk: four LDS reads -- patch rows (ra, rb) x patch columns (ca, cb) -- and
  // V_i = (d[ra][ca] + sb d[ra][cb]) +- (d[rb][ca] + sb d[rb][cb]).  The rows are compile-time (immediate offsets of the reads), the
  // columns wave-uniform (in the four lane bases).  16 more reads per chunk than a column pass shared by the four positions, but
  // nothing is parked in LDS and every position's operand is built INSIDE the MFMA group in front of it (reads of the first half
  // chunk issued ahead of the group, of the second in its middle): the LDS round trips run under the wave's own MFMAs.
  auto rd4 = [&](const float* rawbuf, int i, int kk, f32x4 (&a)[4]) {
    if (DBG & 4) return;
    const int ra = (i == 0) ? 0 : (i == 2 ? 2 : 1), rb = (i == 0 || i == 1) ? 2 : (i == 2 ? 1 : 3);
    a[0] = *reinterpret_cast<const f32x4*>(rawbuf + (ra < 2 ? offa01 : offa23) + ((ra & 1) * VROW + kk * 8));
    a[1] = *reinterpret_cast<const f32x4*>(rawbuf + (ra < 2 ? offb01 : offb23) + ((ra & 1) * VROW + kk * 8));
    a[2] = *reinterpret_cast<const f32x4*>(rawbuf + (rb < 2 ? offa01 : offa23) + ((rb & 1) * VROW + kk * 8));
    a[3] = *reinterpret_cast<const f32x4*>(rawbuf + (rb < 2 ? offb01 : offb23) + ((rb & 1) * VROW + kk * 8));
  };
  auto comb = [&](int i, const f32x4 (&a)[4]) {
    // component by component (and the file is compiled with -fno-slp-vectorize): plain v_fma_f32 / v_sub_f32, which issue beside an
    // MFMA; the packed forms (v_pk_fma_f32) the vector expression compiles to do not (profiles/r05a_mfma_fillers.txt)
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t1 = __builtin_fmaf(a[1][e], sb, a[0][e]), t2 = __builtin_fmaf(a[3][e], sb, a[2][e]);
      r[e] = (i == 1) ? t1 + t2 : t1 - t2;
    }
    return r;
  };
  auto sp3 = [&](const f32x4& lo, const f32x4& hi, bf16x8& h, bf16x8& m, bf16x8& l) {
    if (DBG & 128) {           // ablation: one conversion per value, no residuals
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (__bf16)(e < 4 ? lo[e] : hi[e - 4]);
      m = h; l = h;
      return;
    }
    split3x8(lo, hi, h, m, l);
  };
  bf16x8 vs[3];                   // the split operand of the MFMA group that runs next

  // ---- U fragments: two slots of [nblk][plane] bf16x8, a position (i, j) per slot, straight from global in fragment-major order
  // (k_wino_weights_split); slot s = i & 1 is refilled with position i + 2 right after position i's MFMA group ----
  bf16x8 us[2][2][3];
  const __bf16* ubase_s = nullptr;                  // set per tile (cout block), wave-uniform
  auto load_us = [&](int chunk, int i, int slot) {
    if ((DBG & 16) && chunk != c_begin) return;
    const char* q = reinterpret_cast<const char*>(ubase_s + (size_t)chunk * 16 * (2 * VUS) + (size_t)(4 * i) * (2 * VUS));
    const unsigned vo = (unsigned)lane_now() * 16u;                  // scalar base + 32-bit lane offset (the saddr form of the load)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const char* qn = q + n * (VUS * 2);
      asm volatile("" : "+s"(qn));
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)               // (explicitly global: a laundered pointer would become a flat load)
        us[slot][n][pl] = *(const __attribute__((address_space(1))) bf16x8*)(qn + (size_t)vo + pl * 1024);
    }
  };
  f32x16 acc[4][2];             // [row i][nblk]
  // half of the 12 MFMAs of position (i, j): product-major over the two n blocks (two independent accumulators between dependent
  // MFMAs), smallest terms first
  auto mfma_half = [&](int i, int slot, int half) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    if (DBG & 1) {               // keep the operands live, issue no MFMA
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) acc[i][n][pl + 3 * half] += (float)vs[pl][0] * (float)us[slot][n][pl][0];
      return;
    }
#pragma unroll
    for (int q = 3 * half; q < 3 * half + 3; ++q)
#pragma unroll
      for (int n = 0; n < 2; ++n)
        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vs[PA[q]], us[slot][n][PB[q]], acc[i][n], 0, 0, 0);
  };
  // MFMA group of position i (U in `slot`) with the operand of the NEXT group -- position ni of the raw tile nbuf -- built inside it
  f32x4 ta[4], tlo = {0.f, 0.f, 0.f, 0.f};
  // (DBG & 256, timing only: the operands of the odd positions are not built -- what a V build shared by two groups would cost)
  auto group_pre = [&](const float* nbuf, int ni) { if ((DBG & 256) && (ni & 1)) return; rd4(nbuf, ni, 0, ta); };
  auto group_run = [&](int i, int slot, const float* nbuf, int ni) {
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(i, slot, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (!((DBG & 256) && (ni & 1))) {
      tlo = comb(ni, ta);
      rd4(nbuf, ni, 1, ta);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(i, slot, 1);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto group_post = [&](int ni) {                    // ... and its 3-way split, once the group's MFMAs have read the old operand
    if ((DBG & 256) && (ni & 1)) return;
    const f32x4 thi = comb(ni, ta);
    sp3(tlo, thi, vs[0], vs[1], vs[2]);
  };

  const int T = g.tiles_h * g.tiles_w;                             // statistics partials per image
  const bool stats = direct && p.ostat != nullptr;
  const bool has_res = direct && p.res0 != nullptr;
  const size_t Mtot = (size_t)p.B * H * W;
  float* dst = direct ? p.out : p.partial + (size_t)blockIdx.y * Mtot * p.Cout;

  // ---- first tile: constants, raw chunks 0 and 1 ------------------------------------------------------------------
  int vtile = blockIdx.x;
  int par = 0;
  decode_tile(vtile);
  set_items();
  load_consts();
  const int c1 = min(c_begin + 1, c_end - 1), c2 = min(c_begin + 2, c_end - 1);   // (short K ranges re-fetch their last chunk)
  load_raw(c_begin, rh);
  load_raw(c1, rh2);
  store_consts(cst);
  __syncthreads();

  for (;;) {
    // ================================ prologue ================================
    const float* cs_ = cst + par * VCST_F;
    ubase_s = ufrag + (size_t)cb * nch * 16 * (2 * VUS) + (size_t)wave * (2 * VUS);
    load_us(c_begin, 0, 0);
    load_us(c_begin, 1, 1);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    park_items();
    store_raw(raw0, c_begin, rh, cs_);
    if (nck > 1) store_raw(raw1, c_begin + 1, rh2, cs_);
    load_raw(c2, rh);
    __syncthreads();
    {                                       // operand (chunk 0, position 0)
      rd4(raw0, 0, 0, ta);
      tlo = comb(0, ta);
      rd4(raw0, 0, 1, ta);
      group_post(0);
    }

    // ================================ main loop ================================
    // Per chunk four MFMA groups (rows i = 0..3 of this wave's column).  Inside group i the operand of group i + 1 is built from the
    // raw tile; behind it its U slot is refilled with position i + 2.  The raw tile of chunk c is read from group (c - 1, 3) to group
    // (c, 2); one workgroup barrier per chunk behind group (c, 2): raw[c & 1] is consumed -> chunk c + 2 is staged into it, and chunk
    // c + 1's raw tile (staged a chunk ago) is visible to the reads of group (c, 3).  The last chunk builds an operand nobody uses and
    // re-fetches its own U (straight-line code, unconditional loads).
    for (int i = 0; i < nck; ++i) {
      float* rcur = (i & 1) ? raw1 : raw0;
      const float* rnext = (i & 1) ? raw0 : raw1;
      const bool more = i + 1 < nck;
      const int cc = c_begin + i, cn = c_begin + (more ? i + 1 : i);
      group_pre(rcur, 1);
      group_run(0, 0, rcur, 1);
      load_us(cc, 2, 0);
      group_post(1);
      group_pre(rcur, 2);
      group_run(1, 1, rcur, 2);
      load_us(cc, 3, 1);
      group_post(2);
      group_pre(rcur, 3);
      group_run(2, 0, rcur, 3);
      load_us(cn, 0, 0);
      group_post(3);
      __builtin_amdgcn_sched_barrier(0);
      if (!(DBG & 512)) __syncthreads();             // (DBG & 512, timing only: no barrier in the loop)
      if (i + 2 < nck && !(DBG & 32)) {
        fetch_items();
        store_raw(rcur, c_begin + i + 2, rh, cs_);
        if (i + 3 < nck) load_raw(c_begin + i + 3, rh);
      }
      __builtin_amdgcn_sched_barrier(0);
      group_pre(rnext, 0);
      group_run(3, 1, rnext, 0);
      load_us(cn, 1, 1);
      group_post(0);
    }

    // ================================ epilogue ================================
    // rows folded in registers (A^T = [1 1 1 0; 0 1 -1 -1]):  R_p = sum_i A^T[p][i] M_ij, then Y[p][q] = sum_j R_p(j) A[j][q] through LDS in a
    // FIXED order (bitwise reproducible): q = 0: (R(0) + R(1)) - R'(2), q = 1: (R(1) + R'(2)) - R(3)  (R'(2) = wave 2's sign-flipped plane).
    // Plane (wave j, p) = [32 channels][32 tiles + 4 pad], channels >= 16 shifted by 4 floats: 16-byte writes (lanes = channels) and
    // reads (lanes = channel quads x tile quads) both bank-conflict free; one 32-channel block per round;
    // thread -> (sub-pixel p q, 4 consecutive tiles of a tile row, 4 consecutive channels): 12 reads, a 4 x 4 register transpose,
    // 4 NHWC stores of 16 bytes (8 adjacent lanes = 128 contiguous bytes).
    const int elane = lane_now();
    const int nq = elane & 7, fq = (elane >> 4) & 1, fp = wave & 1;
    const int tq = ((elane >> 5) & 1) | (((elane >> 3) & 1) << 1) | ((wave >> 1) << 2);     // tiles 4 tq .. 4 tq + 3 (half a tile row)
    const int ety = tq >> 1, etx0 = (tq & 1) * 4;
    const int e_cb = cb, e_b0 = b0, e_tix = th_i * g.tiles_w + tw_i, e_vtile = vtile;
    const size_t pix0 = ((size_t)b0 * H + (h0 + 2 * ety + fp)) * W + (w0 + 2 * etx0 + fq);   // tile k of the four: + 2 k pixels
    const bool has_next = vtile + (int)gridDim.x < ntiles;
    if (has_next) vtile += gridDim.x;
    // this tile's residual, both rounds: 8 loads (absent: a valid dummy address, discarded below)
    f32x4 addv[2][4];
#pragma unroll
    for (int nblk = 0; nblk < 2; ++nblk) {
      const int n = e_cb * VBN + nblk * 32 + nq * 4;
      const int ne = n < p.Cout ? n : 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t pix = pix0 + 2 * k;
        const float* rp = dst + pix * p.Cout + ne;
        if (has_res) rp = (ne < p.RC0) ? p.res0 + pix * p.RC0 + ne : p.res1 + pix * p.RC1 + (ne - p.RC0);
        addv[nblk][k] = *reinterpret_cast<const f32x4*>(rp);
      }
    }
    if (DBG & 8) {                 // ablation: no epilogue (one store per thread keeps the accumulators live)
      float s = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) s += acc[a][b][r];
      p.out[(size_t)e_vtile * VNT + tid_now()] = s + addv[0][0][0] + addv[1][3][3];
      __syncthreads();
      decode_tile(vtile);
      load_consts();
      store_consts(cst + (par ^ 1) * VCST_F);
      set_items();
      load_raw(c_begin, rh);
      load_raw(c1, rh2);
      __syncthreads();
      par ^= 1;
      if (!has_next) break;
      continue;
    }
    f32x4 base[2];
#pragma unroll
    for (int nblk = 0; nblk < 2; ++nblk) base[nblk] = *reinterpret_cast<const f32x4*>(cs_ + nblk * 32 + nq * 4);
    __syncthreads();                                                 // the raw tiles and the item table are dead
    float* exch = smem;                                              // [4 waves][2 p][VEPL]
    double s1[2][4], s2[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int k = 0; k < 4; ++k) { s1[a][k] = 0.0; s2[a][k] = 0.0; }
    const int wn = elane & 31;                                       // channel this lane's accumulator column belongs to
    float* wbase = exch + (wave * 2) * VEPL + wn * VETS + (wn >> 4) * 4 + 4 * (elane >> 5);
#pragma unroll
    for (int nblk = 0; nblk < 2; ++nblk) {
      const int n = e_cb * VBN + nblk * 32 + nq * 4;
      const bool nok = n < p.Cout;
      {
        const f32x16 r0 = (acc[0][nblk] + acc[1][nblk]) + acc[2][nblk];
        const f32x16 r1 = (acc[1][nblk] - acc[2][nblk]) - acc[3][nblk];
        // D layout: reg r of lane l -> tile (r & 3) + 4 (l >> 5) of tile row r >> 2, channel l & 31
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          *reinterpret_cast<f32x4*>(wbase + 8 * k) = f32x4{r0[4 * k], r0[4 * k + 1], r0[4 * k + 2], r0[4 * k + 3]};
          *reinterpret_cast<f32x4*>(wbase + VEPL + 8 * k) = f32x4{r1[4 * k], r1[4 * k + 1], r1[4 * k + 2], r1[4 * k + 3]};
        }
      }
      if (nblk == 0) {
        // half of the accumulators are dead: the next tile's constants (4 loads) and raw chunks 0 and 1 are put in flight
        decode_tile(vtile);
        load_consts();
        set_items();
        load_raw(c_begin, rh);
        load_raw(c1, rh2);
      }
      __syncthreads();
      {
        f32x4 y[4];                                                  // [channel jj] over the 4 tiles
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int nn = nq * 4 + jj;
          const float* rb_ = exch + fp * VEPL + nn * VETS + (nn >> 4) * 4 + 4 * tq;
          auto rd = [&](int j) { return *reinterpret_cast<const f32x4*>(rb_ + (j * 2) * VEPL); };
          if (fq == 0) { const f32x4 a0 = rd(0), a1 = rd(1), a2 = rd(2); y[jj] = (a0 + a1) - a2; }
          else { const f32x4 a1 = rd(1), a2 = rd(2), a3 = rd(3); y[jj] = (a1 + a2) - a3; }
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
      if (nblk == 0) store_consts(cst + (par ^ 1) * VCST_F);      // ... and the constants go to LDS (the other parity)
      __syncthreads();                                  // every read of the exchange block is complete
    }
    if (stats) {
      // per-channel sums of this tile's outputs in a fixed order: part[e][thread] (e = [sum | sumsq][32-channel block][channel of
      // the quad]), 256 threads each add 16 of the 32 partials that share a channel quad (threads nq, nq + 8, ...), 128 add two
      double* part = reinterpret_cast<double*>(smem);
      const int t_ = tid_now();
#pragma unroll
      for (int nb2 = 0; nb2 < 2; ++nb2)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          part[(nb2 * 4 + k) * VPR + t_] = s1[nb2][k];
          part[(8 + nb2 * 4 + k) * VPR + t_] = s2[nb2][k];
        }
      __syncthreads();
      {
        const int cq = t_ & 7, e = (t_ >> 3) & 15, grp = t_ >> 7;
        double a = 0.0;
#pragma unroll
        for (int s = 0; s < 16; ++s) a += part[e * VPR + (grp * 16 + s) * 8 + cq];
        part[16 * VPR + grp * 128 + e * 8 + cq] = a;
      }
      __syncthreads();
      if (t_ < 128) {
        const int cq = t_ & 7, e = t_ >> 3;             // e = which * 8 + nb2 * 4 + k
        const double a = part[16 * VPR + t_] + part[16 * VPR + 128 + t_];
        const int which = e >> 3, c = ((e >> 2) & 1) * 32 + cq * 4 + (e & 3);
        const int nn = e_cb * VBN + c;
        if (nn < p.Cout) p.ostat[(((size_t)e_b0 * T + e_tix) * p.Cout + nn) * 2 + which] = a;
      }
      __syncthreads();                                  // the parked sums are read: LDS is free for the next tile
    }
    par ^= 1;
    if (!has_next) break;
  }   // tile loop
}

// ---- host -----------------------------------------------------------------------------------------------------
bool wino2_fits(const ConvParams& p) {
  return p.ksize == 3 && p.stride == 1 && p.Ho == (p.Hs << p.ups) && p.Wo == (p.Ws << p.ups) && p.Wo >= 16 && (p.Wo % 16) == 0 &&
         (p.Ho % 8) == 0 && p.drop_thresh == 0;
}

int conv3x3_wino2_forward(const ConvParams& p, const WinoGeom& g, const float* ufrag, hipStream_t st) {
  if (g.NB != 1 || g.TH != 8 || p.drop_thresh != 0) { set_error("conv: the two-workgroup Winograd kernel covers the one-image 8 x 16 tile without dropout"); return SR3_E_UNSUPPORTED; }
  static const int n_cu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n & ~7;               // a multiple of 8 keeps every workgroup's tiles on one XCD's range of the list
  }();
  const long ntiles = wino_workgroups(p, g);
  dim3 grid((unsigned)std::min<long>(ntiles, 2L * (n_cu > 0 ? n_cu : 256)), p.ksplit);       // persistent: two workgroups per CU
  static const int dbg = [] { const char* e = getenv("SR3_WINO_DBG"); return e ? atoi(e) : 0; }();
#ifdef SR3_WINO_ABLATIONS
  // tooling build only: SR3_W2_ONE_PER_CU=1 asks for more LDS than two workgroups can share, i.e. ONE workgroup per CU -- how much of
  // the kernel's time the second, independent workgroup hides
  static const int lds_bytes = [] { const char* e = getenv("SR3_W2_ONE_PER_CU"); return (e && e[0] == '1') ? 100 * 1024 : V_SMEM; }();
#else
  constexpr int lds_bytes = V_SMEM;
#endif
#define SR3_W2_LAUNCH(D)                                                                                              \
  {                                                                                                                   \
    static std::atomic<uint64_t> done{0};                                                                             \
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(k_conv3x3_wino2<D>), 100 * 1024, done)) return rc;      \
    hipLaunchKernelGGL((k_conv3x3_wino2<D>), grid, dim3(VNT), lds_bytes, st, p, g, reinterpret_cast<const __bf16*>(ufrag)); \
  }
  switch (dbg) {
    case 0: SR3_W2_LAUNCH(0) break;
#ifdef SR3_WINO_ABLATIONS
    case 1: SR3_W2_LAUNCH(1) break;
    case 4: SR3_W2_LAUNCH(4) break;
    case 8: SR3_W2_LAUNCH(8) break;
    case 16: SR3_W2_LAUNCH(16) break;
    case 32: SR3_W2_LAUNCH(32) break;
    case 128: SR3_W2_LAUNCH(128) break;
    case 180: SR3_W2_LAUNCH(180) break;        // 4 + 16 + 32 + 128: the bare MFMA loop + prologue / epilogue
    case 181: SR3_W2_LAUNCH(181) break;        // ... without the MFMAs: prologue / epilogue only
    case 256: SR3_W2_LAUNCH(256) break;        // operands of the odd positions not built (timing only)
    case 288: SR3_W2_LAUNCH(288) break;        // ... and no staging in the loop
    case 512: SR3_W2_LAUNCH(512) break;        // no workgroup barrier in the loop (timing only)
    case 544: SR3_W2_LAUNCH(544) break;        // ... and no staging
#endif
    default: set_error("conv: SR3_WINO_DBG=%d is not built for the two-workgroup Winograd kernel", dbg); return SR3_E_BADARG;
  }
#undef SR3_W2_LAUNCH
  SR3_LAUNCH_CHECK("k_conv3x3_wino2");
  return SR3_OK;
}

}  // namespace sr3
