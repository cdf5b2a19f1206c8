// 1x1 convolutions (attention qkv / out, model/sr3_modules/unet.py:120-121, and ResnetBlock's res_conv, :102-103) as a
// plain NT GEMM on v_mfma_f32_32x32x2_f32:   out[m][n] = sum_c act(A[m][c]) * W[n][c],  m = pixel, n = output channel.
//
// The im2col kernel (conv_igemm.hip) stages BOTH operands through LDS every 32-channel step and reaches 58-83 TF on these
// shapes.  Here only the activations go through LDS: the weights are static between optimizer steps, so they are kept in
// the plan's derived buffer in *fragment-major* order (the trick of conv3x3_wino.hip) -- a wave's B operand for four
// consecutive k-steps is one fully coalesced 1 KB load straight into registers, prefetched one chunk ahead.
// Workgroup = 8 waves (two per CU), tile 128 pixels x 128 output channels; wave (wm, wn) owns rows 64 wm .. +64 and columns
// 32 wn .. +32 (two 32x32 MFMA tiles): per 8-channel group 2 A fragment reads for 8 MFMAs, B from registers.
// Prologue (GroupNorm affine for qkv, none for res_conv / out), concat seam, epilogue (bias, FiLM, residual, 16-byte NHWC
// stores, split-K slabs) as in the other conv kernels; GroupNorm statistics of the output are left to the stand-alone
// pass / the split-K reduce, exactly as for the im2col kernel.
//
// STATUS (round 2): correct (tests/test_gpu_ops.py::test_conv with tile 12, full-network parity with the option on) but
// NOT faster on this network: 68 TF over the 30 1x1 launches of a C2 forward against 72 TF for the im2col kernel's 64x64
// tile, plus more split-K reduces (the 128x128 tile leaves the 16x16 layers with 128 workgroups) -- only 32 MFMAs per
// wave between barriers and K of 64..512 leave it prologue / epilogue bound.  Plan option "gemm1x1", default 0.
#include <stdlib.h>

#include "sr3_common.h"

namespace sr3 {

namespace {
constexpr int GBM = 128, GBN = 128, GBK = 32, GLD = 36, GNT = 512;
constexpr int G_SMEM = 2 * GBM * GLD * 4;          // double-buffered A tile; the epilogue's 8 x [32][36] transposes fit in it

__device__ __forceinline__ float silu_g(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
}
}  // namespace

// W [Cout][Cin] -> gfrag[cout_blk 128][chunk 32][wn 4][kk 4][lane 64][4]:
//   lane l of fragment (wn, kk) holds W[n = blk*128 + wn*32 + (l & 31)][c = chunk*32 + kk*8 + (l >> 5)*4 .. +3], zero outside
__global__ __launch_bounds__(256) void k_gemm_weights(const float* __restrict__ w, int Cout, int Cin, int nch, int ncb,
                                                       float* __restrict__ gfrag) {
  const int quads = nch * 8;
  const long total = (long)ncb * GBN * quads;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(idx % quads);
    const int n = (int)(idx / quads);
    const int c = cq * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < Cout && c < Cin) v = *reinterpret_cast<const f32x4*>(w + (size_t)n * Cin + c);
    const int cb = n / GBN, nl = n - cb * GBN;
    const int wn = nl >> 5;
    const int chunk = c / GBK, cl = c - chunk * GBK;
    const int kk = cl >> 3, hi = (cl >> 2) & 1;
    const int lane = (nl & 31) + 32 * hi;
    *reinterpret_cast<f32x4*>(gfrag + ((size_t)(cb * nch + chunk) * 16 + wn * 4 + kk) * 256 + lane * 4) = v;
  }
}

size_t gemm1x1_weight_floats(int Cout, int Cin) {
  const size_t ncb = (Cout + GBN - 1) / GBN, nch = (Cin + GBK - 1) / GBK;
  return ncb * nch * 16 * 256;
}

int gemm1x1_transform_weights(const float* w, int Cout, int Cin, float* gfrag, hipStream_t st) {
  if (Cin & 3) { set_error("gemm1x1: Cin %% 4 != 0"); return SR3_E_UNSUPPORTED; }
  const int ncb = (Cout + GBN - 1) / GBN, nch = (Cin + GBK - 1) / GBK;
  const long total = (long)ncb * GBN * nch * 8;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_gemm_weights, dim3(blocks), dim3(256), 0, st, w, Cout, Cin, nch, ncb, gfrag);
  SR3_LAUNCH_CHECK("k_gemm_weights");
  return SR3_OK;
}

__global__ __launch_bounds__(GNT, 2) void k_gemm1x1(const ConvParams p, const float* __restrict__ gfrag) {
  extern __shared__ f32x4 smem_v[];
  float* smem = reinterpret_cast<float*>(smem_v);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int Cin = p.C0 + p.C1;
  const int HW = p.Ho * p.Wo;
  const int M = p.B * HW;
  const int tiles_n = (p.Cout + GBN - 1) / GBN;
  const int tile_m = blockIdx.x / tiles_n;
  const int tile_n = blockIdx.x - tile_m * tiles_n;
  const int nch = (Cin + GBK - 1) / GBK;
  const int cper = (nch + p.ksplit - 1) / p.ksplit;
  const int c_begin = blockIdx.y * cper;
  const int c_end = min(nch, c_begin + cper);

  // ---- A loader: row (tid >> 3) + 64 i, channel quad tid & 7 of the chunk ----
  const int kq = tid & 7, lrow = tid >> 3;
  int rm[2], rbimg[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = tile_m * GBM + lrow + 64 * i;
    rm[i] = m < M ? m : -1;
    rbimg[i] = m < M ? m / HW : 0;
  }
  f32x4 ra[2], ssa[2], ssb[2];
  bool cvalid = false;
  auto load_a = [&](int chunk) {
    const int c = chunk * GBK + kq * 4;
    cvalid = c < Cin;
    const int ce = cvalid ? c : 0;
    const bool second = ce >= p.C0;
    const float* sp = second ? p.src1 : p.src0;
    const int sC = second ? p.C1 : p.C0;
    const int cs = second ? ce - p.C0 : ce;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = rm[i] >= 0 ? rm[i] * sC + cs : 0;
      ra[i] = *reinterpret_cast<const f32x4*>(sp + off);
      if (p.act != 0) {
        const float* q = p.ss + ((size_t)rbimg[i] * Cin + ce) * 2;
        ssa[i] = *reinterpret_cast<const f32x4*>(q);
        ssb[i] = *reinterpret_cast<const f32x4*>(q + 4);
      }
    }
  };
  auto store_a = [&](int stage) {
    float* A = smem + stage * (GBM * GLD);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 v = ra[i];
      if (p.act != 0) {
        v.x = fmaf(v.x, ssa[i].x, ssa[i].y);
        v.y = fmaf(v.y, ssa[i].z, ssa[i].w);
        v.z = fmaf(v.z, ssb[i].x, ssb[i].y);
        v.w = fmaf(v.w, ssb[i].z, ssb[i].w);
        if (p.act == 2) { v.x = silu_g(v.x); v.y = silu_g(v.y); v.z = silu_g(v.z); v.w = silu_g(v.w); }
      }
      v = (cvalid && rm[i] >= 0) ? v : zero;
      *reinterpret_cast<f32x4*>(&A[(lrow + 64 * i) * GLD + kq * 4]) = v;
    }
  };

  // ---- B fragments from global, fragment-major ----
  f32x4 u[4], un[4];
  const float* ubase = gfrag + ((size_t)tile_n * nch * 16 + wn * 4) * 256 + lane * 4;
  auto load_u = [&](int chunk, f32x4 (&dst)[4]) {
    const float* q = ubase + (size_t)chunk * 16 * 256;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) dst[kk] = *reinterpret_cast<const f32x4*>(q + kk * 256);
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int kh = (lane >> 5) * 4;
  const int arow = wm * 64 + (lane & 31);

  if (c_begin < c_end) {
    load_a(c_begin);
    load_u(c_begin, u);
    store_a(0);
    __syncthreads();
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
      const int cur = (chunk - c_begin) & 1;
      const bool more = chunk + 1 < c_end;
      if (more) { load_a(chunk + 1); load_u(chunk + 1, un); }
      const float* A = smem + cur * (GBM * GLD);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        f32x4 a[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(&A[(arow + 32 * i) * GLD + kk * 8 + kh]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], u[kk][q], acc[i], 0, 0, 0);
      }
      if (more) {
        store_a(cur ^ 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) u[kk] = un[kk];
      }
      __syncthreads();
    }
  }

  // ---- epilogue: wave-private LDS transpose, 16-byte bias / FiLM / residual / store ----
  // D layout: reg r of lane l -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31
  __syncthreads();
  float* tr = smem + wave * (32 * GLD);
  const bool direct = p.ksplit == 1;
  float* dst = direct ? p.out : p.partial + (size_t)blockIdx.y * M * p.Cout;
  const int c4 = lane & 7;
  const int n = tile_n * GBN + wn * 32 + c4 * 4;
  const bool nok = n < p.Cout;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (direct && nok && p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * GLD + (lane & 31)] = acc[i][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-private region: LDS runs a wave's instructions in order
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = (lane >> 3) + 8 * e;
      const int m = tile_m * GBM + wm * 64 + i * 32 + row;
      f32x4 v = *reinterpret_cast<const f32x4*>(&tr[row * GLD + c4 * 4]);
      if (m < M && nok) {
        if (direct) {
          v += bias4;
          if (p.film) v += *reinterpret_cast<const f32x4*>(p.film + (size_t)(m / HW) * p.film_stride + n);
          if (p.res0) {
            if (n < p.RC0) v += *reinterpret_cast<const f32x4*>(p.res0 + (size_t)m * p.RC0 + n);
            else v += *reinterpret_cast<const f32x4*>(p.res1 + (size_t)m * p.RC1 + (n - p.RC0));
          }
        }
        *reinterpret_cast<f32x4*>(dst + (size_t)m * p.Cout + n) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the reads above are done before the next block overwrites tr
  }
}

bool gemm1x1_fits(const ConvParams& p) {
  return p.ksize == 1 && p.stride == 1 && p.ups == 0 && p.Ho == p.Hs && p.Wo == p.Ws && !p.x2_w && p.drop_thresh == 0;
}
long gemm1x1_workgroups(const ConvParams& p) {
  return (long)(((long)p.B * p.Ho * p.Wo + GBM - 1) / GBM) * ((p.Cout + GBN - 1) / GBN);
}
int gemm1x1_chunks(const ConvParams& p) { return (p.C0 + p.C1 + GBK - 1) / GBK; }

int gemm1x1_forward(const ConvParams& p, const float* gfrag, hipStream_t st) {
  if (!gemm1x1_fits(p)) { set_error("conv: the 1x1 GEMM kernel does not fit this problem"); return SR3_E_UNSUPPORTED; }
  if (!gfrag) { set_error("conv: the 1x1 GEMM kernel needs the fragment-major weights"); return SR3_E_BADARG; }
  if (p.ostat && p.ksplit == 1) { set_error("conv: the 1x1 GEMM kernel does not fuse output statistics"); return SR3_E_UNSUPPORTED; }
  const int nch = gemm1x1_chunks(p);
  if (p.ksplit > 1 && (long)(p.ksplit - 1) * ((nch + p.ksplit - 1) / p.ksplit) >= nch) { set_error("conv: ksplit %d leaves an empty split over %d chunks", p.ksplit, nch); return SR3_E_BADARG; }
  dim3 grid((unsigned)gemm1x1_workgroups(p), p.ksplit);
  hipLaunchKernelGGL(k_gemm1x1, grid, dim3(GNT), G_SMEM, st, p, gfrag);
  SR3_LAUNCH_CHECK("k_gemm1x1");
  return SR3_OK;
}

}  // namespace sr3
