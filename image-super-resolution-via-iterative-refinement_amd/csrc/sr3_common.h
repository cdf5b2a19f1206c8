// Internal shared declarations for libsr3_mi355x (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <atomic>

#include "../../include/sr3_mi355x.h"

namespace sr3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// x = h + m + l with three bf16 terms (8 + 8 + 8 significant bits): each residual is exact in fp32.  A pair at a time: one
// v_cvt_pk_bf16_f32 per plane; the residual x - h of either element is one v_dot2c_f32_bf16 (acc += h.lo * -1 + h.hi * 0 and
// the mirrored constant: the bf16 pair is consumed as it is, no unpack to fp32; the result is exactly representable, so the
// instruction's internal rounding does not matter) -- 7 VALU instructions per pair instead of 9 (-DSR3_SPLIT_NO_DOT2: the
// and / shift / packed-subtract form, A/B builds)
__device__ __forceinline__ void split3_pair(float x0, float x1, bf16x2& h, bf16x2& m, bf16x2& l) {
#ifdef SR3_SPLIT_NO_DOT2
  h[0] = (__bf16)x0; h[1] = (__bf16)x1;
  const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
  m[0] = (__bf16)r0; m[1] = (__bf16)r1;
  const float q0 = r0 - (float)m[0], q1 = r1 - (float)m[1];
  l[0] = (__bf16)q0; l[1] = (__bf16)q1;
#else
  // the selectors (-1, 0) / (0, -1) live in registers: as compile-time constants the compiler folds them into the inline
  // constant -1.0, which this instruction reads as 0xBF800000 = (0, -1) for BOTH (measured: garbage results)
  unsigned s_lo = 0x0000BF80u, s_hi = 0xBF800000u;
  asm("" : "+v"(s_lo), "+v"(s_hi));              // (not volatile: one materialisation per kernel, hoisted out of the loops)
  const bf16x2 sel_lo = __builtin_bit_cast(bf16x2, s_lo), sel_hi = __builtin_bit_cast(bf16x2, s_hi);
  h[0] = (__bf16)x0; h[1] = (__bf16)x1;
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(h, sel_lo, x0, false);
  const float r1 = __builtin_amdgcn_fdot2_f32_bf16(h, sel_hi, x1, false);
  m[0] = (__bf16)r0; m[1] = (__bf16)r1;
  const float q0 = __builtin_amdgcn_fdot2_f32_bf16(m, sel_lo, r0, false);
  const float q1 = __builtin_amdgcn_fdot2_f32_bf16(m, sel_hi, r1, false);
  l[0] = (__bf16)q0; l[1] = (__bf16)q1;
#endif
}
__device__ __forceinline__ void split3(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
  for (int e = 0; e < 4; e += 2) {
    bf16x2 hh, mm, ll;
    split3_pair(v[e], v[e + 1], hh, mm, ll);
    h[e] = hh[0]; h[e + 1] = hh[1]; m[e] = mm[0]; m[e + 1] = mm[1]; l[e] = ll[0]; l[e + 1] = ll[1];
  }
}

// ... for the eight values a lane contributes to one v_mfma_f32_32x32x16_bf16 operand (k = 0..3 from lo, 4..7 from hi)
__device__ __forceinline__ void split3x8(const f32x4& lo, const f32x4& hi, bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    bf16x2 hh, mm, ll;
    split3_pair(e < 4 ? lo[e] : hi[e - 4], e < 4 ? lo[e + 1] : hi[e - 3], hh, mm, ll);
    h[e] = hh[0]; h[e + 1] = hh[1]; m[e] = mm[0]; m[e + 1] = mm[1]; l[e] = ll[0]; l[e + 1] = ll[1];
  }
}
// six bf16 MFMA products of the split operands (a = a_h + a_m + a_l, b likewise), smallest terms first: a.b up to terms <= 2^-24 of it
__device__ __forceinline__ void mfma_split6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16& acc) {
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
  for (int q = 0; q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]], b[PB[q]], acc, 0, 0, 0);
}

// sigmoid / SiLU of the staging and activation-backward steps: libm expf (1 ulp) and the hardware reciprocal v_rcp_f32
// (1 ulp).  Measured against float64 autograd at batch 64 (profiles/r04_grad_probe.txt): parameter gradients are within 5e-7
// of float64 with these, the same as with an IEEE division (-DSR3_EXACT_ACT, A/B builds only) and as stock PyTorch-ROCm.
#ifdef SR3_EXACT_ACT
#define SR3_SIGMOID(v) (1.0f / (1.0f + expf(-(v))))
#define SR3_SILU(v) ((v) / (1.0f + expf(-(v))))
#else
#define SR3_SIGMOID(v) __builtin_amdgcn_rcpf(1.0f + expf(-(v)))
#define SR3_SILU(v) ((v) * __builtin_amdgcn_rcpf(1.0f + expf(-(v))))
#endif

// ---- error plumbing (thread-local message, int codes: 0 ok, >0 hipError_t, <0 engine) ----
enum { SR3_OK = 0 };   // SR3_E_* come from the public header
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
#define SR3_HIP(call)                                   \
  do {                                                  \
    hipError_t _e = (call);                             \
    if (_e != hipSuccess) return sr3::hip_fail(_e, #call); \
  } while (0)
#define SR3_LAUNCH_CHECK(name)                          \
  do {                                                  \
    hipError_t _e = hipGetLastError();                  \
    if (_e != hipSuccess) return sr3::hip_fail(_e, name); \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: remember which devices a kernel instantiation has
// been prepared on (one bit per device ordinal) so a process that drives several GPUs -- the reference's single-process
// DataParallel layout -- sets it on each of them.  `done` is a function-local static of the launcher.
inline int ensure_max_lds(const void* kern, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  SR3_HIP(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return SR3_OK;
  SR3_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.fetch_or(bit, std::memory_order_release);
  return SR3_OK;
}

// The C ABI carries fp32 hyper-parameters (lr, betas, dropout p) while the reference evaluates expressions of them as
// Python floats (double): recover the decimal the caller meant (0.2f -> 0.2, 1e-4f -> 1e-4) = the shortest decimal
// that rounds back to the same float, so derived quantities (Adam bias corrections, the dropout threshold p * 2^32)
// match the reference's double arithmetic exactly.
double meant_double(float f);
// dropout p -> (keep threshold on the 32-bit hash, 1 / (1 - p)); one definition for every entry point and the oracle
inline void dropout_consts(float p, unsigned* thresh, float* scale) {
  const double pd = meant_double(p);
  *thresh = p > 0.f ? (unsigned)(pd * 4294967296.0) : 0u;
  *scale = p > 0.f ? (float)(1.0 / (1.0 - pd)) : 1.0f;
}

// The GroupNorm fold of the op that FOLLOWS, done in the tail of the kernel that completes this tensor (round 6; plan option fold_fuse):
// k_rows_fold (small_kernels.hip) walks the tensor per (image, consumer group) -- as the split-K reduce of a conv (ConvParams::fold)
// or as the stand-alone statistics pass -- so a workgroup holds the whole-image sums of its group's channels: it writes them as this
// tensor's partials (one per image: T = 1), adds the other concat source's channels of the group from ITS partials, and writes the
// consumer's (scale, shift) pairs.  No atomics, fixed summation order; one launch instead of two (reduce / statistics + fold).
struct FoldTail {
  int groups, Ctot, c_off;     // the consumer's GroupNorm: `groups` over Ctot channels; this tensor is channels [c_off, c_off + C) of them
  const double* ostat;         // the other concat source's partials [B][oT][oC][2], its channels at [o_off, o_off + oC); null: single source
  int oC, oT, o_off;
  const float* gamma;          // [Ctot]
  const float* beta;
  float eps;
  float* ss;                   // out: [B][Ctot][2]
  float* mr;                   // out: [B][groups][2] (mean, rstd) or null (training keeps them for the backward)
};
bool fold_tail_fits(int C, int Ctot, int c_off, int groups);
// stand-alone form: statistics of x [B][HW][C] (partials [B][1][C][2] into `stat`) + the fold
int chan_stats_fold(const float* x, int B, int HW, int C, double* stat, const FoldTail& f, hipStream_t st);
struct ConvParams;
int splitk_reduce_fold(const ConvParams& p, const FoldTail& f, hipStream_t st);

// ---- convolution (implicit GEMM on v_mfma_f32_32x32x2_f32) --------------------------------
// Activations are NHWC fp32.  The input is the *virtual* channel concat of up to two sources,
// optionally nearest-upsampled x2 (gather in the address math), optionally strided.
// Prologue (per input element, before zero padding): none | x*scale+shift | silu(x*scale+shift)
// with (scale, shift) per (image, input channel) = GroupNorm folded with its affine.
// Epilogue: + bias[n] + film[b][n] + residual[m][n] (residual is itself a two-source concat view).
struct ConvParams {
  const float* src0;
  const float* src1;
  int C0, C1;          // channels of src0 / src1 (C1 = 0 when no concat); both % 4 == 0
  int B, Hs, Ws;       // source spatial dims
  int ups;             // 1: nearest x2 upsample of the source before the conv
  int stride;          // 1 | 2
  int ksize;           // 1 | 3  (pad = ksize / 2)
  int Ho, Wo;          // output spatial dims
  int Cout;
  const float* w;      // [Cout][ksize*ksize][Cin]   (OHWI)
  const float* bias;   // [Cout] or null
  const float* ss;     // [B][Cin][2] (scale, shift) or null
  int act;             // 0 none, 1 affine, 2 affine + SiLU
  const float* film;   // film[b * film_stride + n] or null
  int film_stride;
  const float* res0;   // residual, NHWC [B,Ho,Wo,RC0 (+RC1)] or null
  const float* res1;
  int RC0, RC1;
  float* out;          // [B,Ho,Wo,Cout]
  float* partial;      // split-K scratch [ksplit][M][Cout] (ksplit > 1)
  int ksplit;
  double* ostat;       // optional (halo kernel only): partial {sum, sumsq} of the OUTPUT, [B][T][Cout][2]
  int dbg;             // profiling ablations only (env SR3_CONV_DBG): 1 = skip MFMA + fragment reads, 2 = skip staging, 4 = skip the global loads only, 8 = skip the LDS writes (and the split) only
  // Optional second K-segment (halo kernel only): a 1x1 conv of another tensor (virtual concat
  // x2_src0|x2_src1, same spatial size as the output, no prologue) accumulated into the same
  // output tile -- ResnetBlock's `res_conv(x)` (unet.py:102-103,110) folded into block2's conv.
  const float* x2_src0;
  const float* x2_src1;
  int x2_C0, x2_C1;
  const float* x2_w;    // [Cout][x2_C0 + x2_C1]
  const float* x2_bias; // [Cout] or null
  // Train-mode dropout between the activation and the conv (nn.Dropout of Block, unet.py:86): an
  // activated input element with NHWC linear index i is kept iff hash32(i * 0x9E3779B9 + drop_seed) >=
  // drop_thresh and scaled by drop_scale = 1 / (1 - p).  drop_thresh == 0 disables it.  Only defined for
  // single-source, non-upsampled inputs (block2's input is never a concat).
  unsigned drop_seed, drop_thresh;
  float drop_scale;
  // tile_cfg 11 (Winograd): this conv's transformed filters in fragment-major order (conv3x3_wino.hip)
  const float* wino_u;
  int wino_split;      // 1: the filters are the 3 x bf16 split form and the kernel's SPLIT instantiation runs (tile_cfg 12 at the ABI);
                       // 2: the same filters on the two-workgroups-per-CU kernel of conv3x3_wino2.hip (tile_cfg 13; 8 x 16 pixel tile)
  int igemm_split;     // im2col kernel (tile_cfg 1-4; 1x1 and stride-2 convs): 1 = its 3 x bf16 split instantiation (tile_cfg 14-17 at the ABI)
  int wgrad_split;     // weight gradient (wgrad.hip): 1 = the one-tap-per-workgroup kernel's 3 x bf16 split instantiation for the layers with > 64 channels on both sides
  const FoldTail* fold; // host side only (conv_forward): split-K convs -- the reduce runs as k_rows_fold and does the next op's GroupNorm fold too;
                        // ostat then holds ONE partial per image
  const void* w_split; // igemm_split: the weights pre-split into bf16 planes by igemm_split_weights (tile_cfg 18-21 at the ABI; a plan
                       // keeps them in its derived buffer); null: the kernel splits the weights while it stages them
};

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float drop_mask(unsigned seed, unsigned idx, unsigned thresh, float scale) {
  return hash32(idx * 0x9E3779B9U + seed) >= thresh ? scale : 0.f;
}

// tile_cfg: 0 = auto; im2col-staged implicit GEMM: 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 64x128 (ConvParams::igemm_split
// selects the 3 x bf16 split instantiation of the same tiles);
// halo-tile 3x3 stride-1 kernel: 5 = 128x128, 6 = 256(M)x64(N).  ksplit: 0 = auto
int conv_forward(const ConvParams& p, int tile_cfg, int ksplit, float* splitk_scratch,
                 size_t splitk_scratch_bytes, hipStream_t st);
// scratch bytes the auto heuristic may ask for (upper bound) for this problem
size_t conv_splitk_bytes(const ConvParams& p, int tile_cfg, int ksplit);
void conv_pick(const ConvParams& p, int& tile_cfg, int& ksplit);
int splitk_rows_per_block(const ConvParams& p, bool stats);
// pre-split weights of the im2col SPLIT instantiations: floats of the derived block for `numel` weights, and the transform
size_t igemm_wsplit_floats(int Cout, int taps, int Cin);
int igemm_split_weights(const float* w, int Cout, int taps, int Cin, float* out, hipStream_t st);
// 1x1 stride-1 convs as a plain GEMM on the split arithmetic (gemm1x1.hip; tile_cfg 22: 64 x 128 tile, mi = 2; needs w_split)
bool gemm1x1_fits(const ConvParams& p, int mi);
int gemm1x1_cols(const ConvParams& p);      // 128, or 64 (waves 2 x 2) where Cout % 128 != 0
int gemm1x1_forward(const ConvParams& p, int mi, hipStream_t st);
int gemm1x1_rows(const ConvParams& p);      // 64, or 32 where 64-row tiles would leave workgroup slots empty
// profiling aid: when non-null, conv_forward records this event between the GEMM kernel and the
// split-K reduce kernel (then resets the pointer).  Thread-local.
void conv_set_mid_event(hipEvent_t ev);
struct HaloGeom {
  int TH, TW, NB;          // spatial tile per image, images per workgroup tile (TH*TW*NB = BM)
  int log_tw, log_thw;     // log2(TW), log2(TH*TW)
  int tiles_w, tiles_h;    // tiles per image
  int HPI, HP;             // halo pixels per image / per workgroup
};
// halo-tile configurations: 5 = 128(M) x 128(N), 6 = 256 x 64 (4 waves, two workgroups per CU); 9 = 256 x 128
// (8 waves, one workgroup per CU); 7 / 8 / 10 = the split-bf16 instantiations of 5 / 6 / 9 (8 runs its 256 x 64
// tile on 8 waves of 64 x 32)
inline bool halo_cfg_one_wg_per_cu(int cfg) { return cfg >= 8; }
inline int halo_cfg_bm(int cfg) { return (cfg == 5 || cfg == 7) ? 128 : 256; }
inline int halo_cfg_bn(int cfg) { return (cfg == 6 || cfg == 8) ? 64 : 128; }
inline bool halo_cfg_split(int cfg) { return cfg == 7 || cfg == 8 || cfg == 10; }
bool halo_geometry(const ConvParams& p, int cfg, HaloGeom* g);
int conv3x3_halo_forward(const ConvParams& p, int cfg, const HaloGeom& g, hipStream_t st);

// Winograd F(2x2, 3x3) form of the same conv (conv3x3_wino.hip): tile_cfg 11.  The transformed filters U = G g G^T live
// in a "derived" buffer in fragment-major order (wino_weight_floats floats per conv), rebuilt by wino_transform_weights
// whenever the weights change.
struct WinoGeom {
  int TH, TW, NB;        // output pixels per image in the workgroup tile, images per tile
  int twt, log_twt;      // Winograd (2x2) tiles per tile row
  int tpi, log_tpi;      // Winograd tiles per image of the workgroup tile
  int tiles_w, tiles_h;  // workgroup tiles per image
  int nbt;               // batch tiles: B (one image per tile) or B / 4 (NB = 4: four 8 x 8 images per tile)
  int HPI, HP;           // raw halo pixels per image / per workgroup
  int log_tw, log_th, pow2;   // tile decode of the persistent kernel: tiles_w / tiles_h as shifts when both are powers of two
  unsigned sp_magic;          // ceil(2^32 / (tiles_w * tiles_h * B)): tile id / tiles-per-cout-block as a multiply-high
};
bool wino_geometry(const ConvParams& p, WinoGeom* g);
int wino_stats_slices(const WinoGeom& g);
int wino_max_chunks_per_split(const WinoGeom& g);     // 64 (1024 input channels); 16 for the four-image tile
long wino_workgroups(const ConvParams& p, const WinoGeom& g);
int wino_chunks(const ConvParams& p);
size_t wino_weight_floats(int Cout, int Cin, bool split = false);
int wino_transform_weights(const float* w_ohwi, int Cout, int Cin, float* ufrag, hipStream_t st, bool split = false);
int conv3x3_wino_forward(const ConvParams& p, const float* ufrag, hipStream_t st);
// the 3 x bf16 split form as two independent four-wave workgroups per CU, 8 x 16 pixel tile (conv3x3_wino2.hip; ConvParams::wino_split == 2,
// tile_cfg 13 at the ABI; wino_geometry gives that tile's geometry when wino_split == 2)
bool wino2_fits(const ConvParams& p);
int conv3x3_wino2_forward(const ConvParams& p, const WinoGeom& g, const float* ufrag, hipStream_t st);

// ---- small kernels ------------------------------------------------------------------------
// partial per-(b, channel) {sum, sumsq} in double of an NHWC tensor [B, HW, C]:
// stat[B][T][C][2] with T = chan_stats_slices(B, HW, C); plain stores, no memset needed.
int chan_stats_slices(int B, int HW, int C);
int chan_stats(const float* x, int B, int HW, int C, double* stat, hipStream_t st);
// GroupNorm fold: partial stats of up to two concat sources + gamma/beta -> ss[B][C0+C1][2]
int gn_finalize(const double* stat0, int C0, int T0, const double* stat1, int C1, int T1, int B, int HW, int groups,
                const float* gamma, const float* beta, float eps, float* ss, hipStream_t st, float* mr = nullptr);
// partials per image the halo conv writes into ConvParams::ostat for this geometry
int halo_stats_slices(const HaloGeom& g);
// first conv of the UNet: NCHW inputs (virtual concat of a: Ca, b: Cb channels), 3x3 pad 1,
// weights OHWI [Cout][9][Ca+Cb], output NHWC [B,H,W,Cout]
// statistics partials per image conv_in_nchw writes into `ostat` for this shape (0: no fused statistics: VALU kernel)
int conv_in_stat_slices(int Cin, int H, int W, int Cout);
int conv_in_nchw(const float* a, int Ca, const float* b, int Cb, int B, int H, int W, const float* w,
                 const float* bias, int Cout, float* out, double* ostat, hipStream_t st);
// fused reverse-step update (NCHW, elementwise): coef = {a, b, c1, c2, sigma} tables of length T
struct StepTables { const float* a; const float* b; const float* c1; const float* c2; const float* sigma; };
// the tail of one reverse step folded into the output conv's epilogue (sr3_reverse_step): x <- p_sample update(x, eps, z, t) with
// t = *step_cur, the same separately rounded operations as k_p_sample_update (bit-identical), and *step_next = t - 1 (one thread)
struct StepFuse {
  float* x;               // [B, Cout, H, W] in / out
  const float* z;         // noise or null (= 0)
  StepTables tb;
  const int* step_cur;    // t of this step (written by the embedding kernel of the same forward)
  int* step_next;         // t of the next step
  int clip;
};
// final Block: silu(gn(x)) -> conv3x3 C->Cout(<=4), NHWC in, NCHW out (out_nchw may be null when `fuse` is given)
int conv_out_nchw(const float* x, const float* ss, int B, int H, int W, int C, const float* w,
                  const float* bias, int Cout, float* out_nchw, hipStream_t st, const StepFuse* fuse = nullptr);
// noise-level / timestep embedding + MLP + all FiLM projections
struct EmbedParams {
  int variant;            // 0 sr3 (continuous level), 1 ddpm (integer t)
  int B, inner;           // embedding dim = inner, hidden = 4*inner
  const float* level;     // [B] (sr3) or null
  const int64_t* tstep;   // [B] (ddpm) or null
  const float* level_table;  // sr3: level = level_table[step_dev[0] + 1] when step_dev != null
  const int* step_dev;    // device step counter (graph replay) or null
  int* step_out;          // optional: block 0 copies *step_dev here (sr3_reverse_step: the slot the step's tail kernel reads)
  const float* freq;      // [inner/2] frequency table
  const float* w1; const float* b1;   // [4*inner][inner], [4*inner]
  const float* w2; const float* b2;   // [inner][4*inner], [inner]
  const float* wf; const float* bf;   // concatenated FiLM projections [F][inner], [F]
  int F;
  float* temb;            // [B][inner] scratch
  float* film;            // [B][F]
};
int embed_forward(const EmbedParams& p, hipStream_t st);
// single-head attention over NHWC qkv [B][N][3C] -> out [B][N][C]
// split != 0: the staging-free kernel's 3 x bf16 split instantiation where the shape takes that kernel (fp32 MFMA otherwise)
int attention_forward(const float* qkv, int B, int N, int C, float* out, hipStream_t st, int split = 0);
int p_sample_update(float* x, const float* eps, const float* z, StepTables tb, const int* step_dev,
                    const int64_t* t_per_sample, int step_host, int B, int per_image, hipStream_t st, bool clip = true);
int step_decrement(int* step_dev, hipStream_t st);
// split3_pair self-test (small_kernels.hip): *bad_dev = number of elements whose three bf16 terms do not add back exactly
int split3_selftest(int* bad_dev, hipStream_t st);
// q_sample (sr3: per-sample gamma; ddpm: a[t], s[t]) -> x_noisy ; l1 loss sum
int q_sample(const float* x0, const float* z, const float* ca, const float* cb, int B, int per_image,
             float* out, hipStream_t st);

}  // namespace sr3
