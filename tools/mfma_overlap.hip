// Micro-benchmark: does work from the OTHER wave of a SIMD overlap v_mfma_f32_32x32x2_f32 on gfx950?
// A 512-thread workgroup puts two waves on each SIMD.  Waves 0-3 ("matrix role") run a bare MFMA stream (4 independent
// accumulators); waves 4-7 ("partner role") run one of: nothing, plain v_fma_f32 chains, v_exp_f32, ds_read_b128, global loads.
// Reported: time of each role alone and of both together -- together ~ max(alone) means the pipes overlap across waves,
// together ~ sum means they serialise.  Second part: the same VALU work placed INSIDE the MFMA wave (between MFMAs).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { P_NONE, P_FMA, P_EXP, P_LDS, P_GLD };

template <int PARTNER, bool MATRIX, int INNER>
__global__ __launch_bounds__(512, 1) void k(float* out, const float* src, int iters) {
  __shared__ f32x4 buf[2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  buf[tid] = f32x4{1.f, 2.f, 3.f, 4.f};
  buf[tid + 512] = f32x4{1.f, 2.f, 3.f, 4.f};
  buf[tid + 1024] = f32x4{1.f, 2.f, 3.f, 4.f};
  buf[tid + 1536] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  float s = 0.f;
  if (wave < 4) {
    if (MATRIX) {
      f32x16 acc[4];
      for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      f32x4 a = buf[lane], b = buf[lane + 256];
      float v0 = a.x, v1 = a.y, v2 = a.z, v3 = a.w;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[(q + i) & 3], acc[i], 0, 0, 0);
            if (INNER > 0) {       // INNER plain FMAs per MFMA inside the matrix wave itself
#pragma unroll
              for (int u = 0; u < INNER; ++u) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(1.0001f), "v"(0.5f));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(1.0001f), "v"(0.5f));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(1.0001f), "v"(0.5f));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(1.0001f), "v"(0.5f));
              }
            }
          }
      }
      for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
      s += v0 + v1 + v2 + v3;
    }
  } else {
    if (PARTNER == P_FMA) {
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = buf[lane][i & 3] + i;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u)          // 256 independent-ish FMAs per iteration (= per 16 partner MFMAs)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(1.0001f), "v"(0.5f));
      }
      for (int i = 0; i < 8; ++i) s += v[i];
    } else if (PARTNER == P_EXP) {
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = buf[lane][i & 3] * 0.01f + i * 0.001f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)           // 64 v_exp_f32 per iteration
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      }
      for (int i = 0; i < 8; ++i) s += v[i];
    } else if (PARTNER == P_LDS) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) a += buf[(lane + 64 * u + it * 7) & 2047];     // 32 ds_read_b128 per iteration
      }
      s = a.x + a.y + a.z + a.w;
    } else if (PARTNER == P_GLD) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      const f32x4* g = reinterpret_cast<const f32x4*>(src) + (size_t)blockIdx.x * 4096;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a += g[(lane + 64 * u + 512 * (wave - 4) + it * 64) & 4095];   // 8 x 1 KB per iteration
      }
      s = a.x + a.y + a.z + a.w;
    }
  }
  out[blockIdx.x * 512 + tid] = s;
}

template <int PARTNER, bool MATRIX, int INNER>
float run(float* out, const float* src) {
  const int iters = 2048, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PARTNER, MATRIX, INNER>), dim3(blocks), dim3(512), 0, 0, out, src, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<PARTNER, MATRIX, INNER>), dim3(blocks), dim3(512), 0, 0, out, src, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *out, *src;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&src, 256 * 4096 * 16);
  hipMemset(src, 0, 256 * 4096 * 16);
  const float m = run<P_NONE, true, 0>(out, src);
  const double fl = 256.0 * 4 * 2048 * 16 * 4096.0;
  printf("matrix role alone (1 wave/SIMD busy)      %.3f ms  %.1f TFLOP/s\n", m, fl / m / 1e9);
  printf("partner role, alone / with the matrix role (ms):\n");
  printf("  256 v_fma_f32 per 16 MFMA   alone %.3f   both %.3f\n", run<P_FMA, false, 0>(out, src), run<P_FMA, true, 0>(out, src));
  printf("  64 v_exp_f32 per 16 MFMA    alone %.3f   both %.3f\n", run<P_EXP, false, 0>(out, src), run<P_EXP, true, 0>(out, src));
  printf("  32 ds_read_b128 per 16 MFMA alone %.3f   both %.3f\n", run<P_LDS, false, 0>(out, src), run<P_LDS, true, 0>(out, src));
  printf("  8 global 1KB loads per 16   alone %.3f   both %.3f\n", run<P_GLD, false, 0>(out, src), run<P_GLD, true, 0>(out, src));
  printf("VALU inside the matrix wave (4 x INNER v_fma per MFMA), ms:  0: %.3f  1: %.3f  2: %.3f  4: %.3f\n", m,
         run<P_NONE, true, 1>(out, src), run<P_NONE, true, 2>(out, src), run<P_NONE, true, 4>(out, src));
  return 0;
}
