// Micro-benchmark (round 5): what one v_mfma_f32_32x32x16_bf16 gap hides, per filler class, on gfx950.
//
// Round 4's table (tools/mfma_overlap_bf16.hip) left two questions open: (a) its in-wave row mixed builtin MFMAs with
// `asm volatile` fillers, so the compiler was free to cluster the MFMAs, and every filler was a dependent chain; (b) it priced one
// filler class (v_fma_f32).  Here every instruction of the stream is `asm volatile` (program order = source order), fillers rotate
// over eight independent registers, and the same multiset is timed in three placements:
//   interleaved  each MFMA followed by its N fillers                       (what a hand-scheduled loop would do)
//   clustered    16 MFMAs, then the 16 N fillers                           (what the shipped 8-wave Winograd loop does: MFMA groups
//                                                                           fenced from the VALU sections by sched_barrier)
//   anti-phase   clustered, the second wave of every SIMD starts with the filler cluster (two waves per SIMD only)
// with one wave per SIMD (256 threads) and two (512 threads, both waves run the same stream).
// Output: shader cycles per MFMA slot (= 32 x time / time of the bare one-wave stream), so 32.0 = everything hidden.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_fillers.hip -o /tmp/mfma_fillers && /tmp/mfma_fillers
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { C_FMA, C_SUB, C_CVT, C_PKFMA, C_PKADD, C_DOT2, C_EXP, C_MOV, C_AND, C_CNDMASK, C_LDSR, C_MOV64, C_NCLS };
static const char* kNames[C_NCLS] = {"v_fma_f32", "v_sub_f32", "v_cvt_pk_bf16_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_dot2c_f32_bf16",
                                     "v_exp_f32", "v_mov_b32", "v_and_b32", "v_cndmask_b32", "ds_read_b128", "v_mov_b64"};

#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(ab), "v"(bb))

template <int CLS>
__device__ __forceinline__ void filler(float (&v)[8], f32x2 (&w)[4], f32x4& ld, unsigned addr, int i) {
  float& x = v[i & 7];
  const float y = v[(i + 3) & 7];
  if (CLS == C_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(y));
  else if (CLS == C_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  else if (CLS == C_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  else if (CLS == C_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[i & 3]) : "v"(w[(i + 1) & 3]));
  else if (CLS == C_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[i & 3]) : "v"(w[(i + 1) & 3]));
  else if (CLS == C_DOT2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x) : "v"(y), "v"(v[(i + 5) & 7]));
  else if (CLS == C_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  else if (CLS == C_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
  else if (CLS == C_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(y));
  else if (CLS == C_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y));
  else if (CLS == C_LDSR) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld) : "v"(addr), "n"(0));
  else if (CLS == C_MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(w[i & 3]) : "v"(w[(i + 1) & 3]));
}

// MODE 0 interleaved, 1 clustered, 2 anti-phase (waves >= 4 run the filler cluster first), 3 ping-pong: clustered with an
// s_barrier behind every cluster, waves >= 4 one barrier ahead -- the workgroup barrier is the metronome that keeps one wave of
// every SIMD in its MFMA cluster while the other is in its filler cluster (MT MFMAs per cluster)
template <int CLS, int N, int MODE, int MT = 16>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  __shared__ f32x4 buf[2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2048; i += blockDim.x) buf[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x4 a = buf[lane], b = buf[lane + 256];
  bf16x8 ab = *reinterpret_cast<bf16x8*>(&a), bb = *reinterpret_cast<bf16x8*>(&b);
  float v[8];
  f32x2 w[4];
  for (int i = 0; i < 8; ++i) v[i] = buf[lane + 64 * i][i & 3] * 1e-3f;
  for (int i = 0; i < 4; ++i) w[i] = f32x2{v[i], v[i + 4]};
  f32x4 ld = a;
  const unsigned addr = (unsigned)(lane * 16);
  const bool second = MODE == 2 && wave >= 4;
  if (second) {           // its filler cluster first: from here on the two waves of a SIMD alternate
#pragma unroll
    for (int i = 0; i < 16 * N; ++i) filler<CLS>(v, w, ld, addr, i);
  }
  if (MODE == 3 && wave >= 4) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) {
#pragma unroll
      for (int q = 0; q < MT; ++q) MFMA(acc[q & 3]);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < MT * N; ++i) filler<CLS>(v, w, ld, addr, i);
      __builtin_amdgcn_s_barrier();
    } else if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        MFMA(acc[q & 3]);
#pragma unroll
        for (int u = 0; u < N; ++u) filler<CLS>(v, w, ld, addr, q * N + u);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) MFMA(acc[q & 3]);
#pragma unroll
      for (int i = 0; i < 16 * N; ++i) filler<CLS>(v, w, ld, addr, i);
    }
    if (CLS == C_LDSR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (MODE == 3 && wave < 4) __builtin_amdgcn_s_barrier();
  float s = ld.x;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += w[i].x + w[i].y;
  out[blockIdx.x * 512 + tid] = s;
}

template <int CLS, int N, int MODE, int MT = 16>
float run(float* out, int threads) {
  const int iters = 2048 * 16 / MT, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CLS, N, MODE, MT>), dim3(blocks), dim3(threads), 0, 0, out, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CLS, N, MODE, MT>), dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms;
}

static float g_bare = 1.f;
template <int CLS>
void row(float* out) {
  // cycles per MFMA slot of ONE wave's stream; two waves per SIMD execute twice the MFMAs, so their figure is per MFMA of the SIMD x 2
  auto cyc = [](float ms) { return 32.f * ms / g_bare; };
  printf("%-18s 1 wave/SIMD interleaved  N=2 %6.1f  N=4 %6.1f  N=5 %6.1f  N=6 %6.1f  N=8 %6.1f\n", kNames[CLS], cyc(run<CLS, 2, 0>(out, 256)),
         cyc(run<CLS, 4, 0>(out, 256)), cyc(run<CLS, 5, 0>(out, 256)), cyc(run<CLS, 6, 0>(out, 256)), cyc(run<CLS, 8, 0>(out, 256)));
  printf("%-18s 1 wave/SIMD clustered    N=2 %6.1f  N=4 %6.1f  N=5 %6.1f  N=6 %6.1f  N=8 %6.1f\n", "", cyc(run<CLS, 2, 1>(out, 256)),
         cyc(run<CLS, 4, 1>(out, 256)), cyc(run<CLS, 5, 1>(out, 256)), cyc(run<CLS, 6, 1>(out, 256)), cyc(run<CLS, 8, 1>(out, 256)));
  printf("%-18s 2 waves/SIMD interleaved N=2 %6.1f  N=4 %6.1f  N=5 %6.1f  N=6 %6.1f  N=8 %6.1f   (per MFMA of the SIMD: / 2)\n", "",
         cyc(run<CLS, 2, 0>(out, 512)), cyc(run<CLS, 4, 0>(out, 512)), cyc(run<CLS, 5, 0>(out, 512)), cyc(run<CLS, 6, 0>(out, 512)),
         cyc(run<CLS, 8, 0>(out, 512)));
  printf("%-18s 2 waves/SIMD clustered   N=2 %6.1f  N=4 %6.1f  N=5 %6.1f  N=6 %6.1f  N=8 %6.1f\n", "", cyc(run<CLS, 2, 1>(out, 512)),
         cyc(run<CLS, 4, 1>(out, 512)), cyc(run<CLS, 5, 1>(out, 512)), cyc(run<CLS, 6, 1>(out, 512)), cyc(run<CLS, 8, 1>(out, 512)));
  printf("%-18s 2 waves/SIMD anti-phase  N=2 %6.1f  N=4 %6.1f  N=5 %6.1f  N=6 %6.1f  N=8 %6.1f\n", "", cyc(run<CLS, 2, 2>(out, 512)),
         cyc(run<CLS, 4, 2>(out, 512)), cyc(run<CLS, 5, 2>(out, 512)), cyc(run<CLS, 6, 2>(out, 512)), cyc(run<CLS, 8, 2>(out, 512)));
  printf("%-18s 2 waves/SIMD ping-pong16 N=2 %6.1f  N=4 %6.1f  N=5 %6.1f  N=6 %6.1f  N=8 %6.1f   (s_barrier behind every 16-MFMA / 16N-filler cluster)\n", "",
         cyc(run<CLS, 2, 3>(out, 512)), cyc(run<CLS, 4, 3>(out, 512)), cyc(run<CLS, 5, 3>(out, 512)), cyc(run<CLS, 6, 3>(out, 512)), cyc(run<CLS, 8, 3>(out, 512)));
  printf("%-18s 2 waves/SIMD ping-pong12 N=2 %6.1f  N=4 %6.1f  N=5 %6.1f  N=6 %6.1f  N=8 %6.1f   (12-MFMA clusters: the Winograd loop's groups)\n", "",
         cyc(run<CLS, 2, 3, 12>(out, 512)), cyc(run<CLS, 4, 3, 12>(out, 512)), cyc(run<CLS, 5, 3, 12>(out, 512)), cyc(run<CLS, 6, 3, 12>(out, 512)), cyc(run<CLS, 8, 3, 12>(out, 512)));
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  g_bare = run<C_FMA, 0, 0>(out, 256);
  const float two = run<C_FMA, 0, 0>(out, 512);
  const double fl = 256.0 * 4 * 2048 * 16 * 2.0 * 32 * 32 * 16;
  printf("bare v_mfma_f32_32x32x16_bf16 stream: one wave per SIMD %.3f ms = %.0f TFLOP/s (32 cycles per MFMA); two waves per SIMD %.3f ms "
         "(%.1f cycles per MFMA of one wave's stream)\n", g_bare, fl / g_bare / 1e9, two, 32.f * two / g_bare);
  printf("cycles per MFMA slot of one wave's stream, N fillers per MFMA:\n");
  row<C_FMA>(out); row<C_SUB>(out); row<C_CVT>(out); row<C_PKFMA>(out); row<C_PKADD>(out); row<C_DOT2>(out);
  row<C_EXP>(out); row<C_MOV>(out); row<C_MOV64>(out); row<C_AND>(out); row<C_CNDMASK>(out); row<C_LDSR>(out);
  return 0;
}
