// Micro-benchmark (round 4): what overlaps v_mfma_f32_32x32x16_bf16 on gfx950 -- the 3 x bf16 split kernels' matrix
// instruction -- from the OTHER wave of the SIMD and from inside the same wave?  Same layout as tools/mfma_overlap.hip (512
// threads = two waves per SIMD, waves 0-3 run a bare MFMA stream over 4 independent accumulators, waves 4-7 a partner role),
// with partner roles that are PURE in their instruction class: VALU only (v_fma_f32 / v_cvt_pk_bf16_f32), LDS only
// (ds_read_b128 at fixed offsets, no address arithmetic, no use of the data), LDS writes only.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap_bf16.hip -o /tmp/mfma_overlap_bf16 && /tmp/mfma_overlap_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { P_NONE, P_FMA, P_CVT, P_LDSR, P_LDSW };

template <int PARTNER, bool MATRIX, int INNER, bool BF16>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  __shared__ f32x4 buf[4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4096; i += 512) buf[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  float s = 0.f;
  if (wave < 4) {
    if (MATRIX) {
      f32x16 acc[4];
      for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      f32x4 a = buf[lane], b = buf[lane + 256];
      bf16x8 ab = *reinterpret_cast<bf16x8*>(&a), bb = *reinterpret_cast<bf16x8*>(&b);
      float v0 = a.x, v1 = a.y, v2 = a.z, v3 = a.w;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (BF16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[(q + i) & 3], acc[i], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < INNER; ++u) {       // INNER plain FMAs per MFMA inside the matrix wave itself
              asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(1.0001f), "v"(0.5f));
              if (u & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(1.0001f), "v"(0.5f));
              else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(1.0001f), "v"(0.5f));
            }
          }
      }
      for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
      s += v0 + v1 + v2 + v3;
    }
  } else {
    if (PARTNER == P_FMA || PARTNER == P_CVT) {
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = buf[lane][i & 3] + i;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)          // 128 VALU instructions per iteration (= per 16 matrix-role MFMAs)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (PARTNER == P_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(1.0001f), "v"(0.5f));
            else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
          }
      }
      for (int i = 0; i < 8; ++i) s += v[i];
    } else if (PARTNER == P_LDSR) {
      const unsigned addr = (unsigned)(lane * 16);
      f32x4 t0, t1, t2, t3;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {          // 16 ds_read_b128 per iteration, nothing else
          asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(t0) : "v"(addr));
          asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(t1) : "v"(addr));
          asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(t2) : "v"(addr));
          asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(t3) : "v"(addr));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      s = t0.x + t1.x + t2.x + t3.x;
    } else if (PARTNER == P_LDSW) {
      const unsigned addr = (unsigned)(tid * 8);
      float2 t = {1.f, 2.f};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {          // 24 ds_write_b64 per iteration
          asm volatile("ds_write_b64 %0, %1 offset:0" :: "v"(addr), "v"(t));
          asm volatile("ds_write_b64 %0, %1 offset:4096" :: "v"(addr), "v"(t));
          asm volatile("ds_write_b64 %0, %1 offset:8192" :: "v"(addr), "v"(t));
          asm volatile("ds_write_b64 %0, %1 offset:12288" :: "v"(addr), "v"(t));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      s = t.x;
    }
  }
  out[blockIdx.x * 512 + tid] = s;
}

template <int PARTNER, bool MATRIX, int INNER, bool BF16>
float run(float* out) {
  const int iters = 4096, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PARTNER, MATRIX, INNER, BF16>), dim3(blocks), dim3(512), 0, 0, out, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<PARTNER, MATRIX, INNER, BF16>), dim3(blocks), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <bool BF16>
void report(float* out, const char* name, double flop_per_mfma) {
  const float m = run<P_NONE, true, 0, BF16>(out);
  const double fl = 256.0 * 4 * 4096 * 16 * flop_per_mfma;
  printf("%s\n  matrix role alone (1 wave per SIMD, 16 MFMAs per iteration)  %.3f ms  %.1f TFLOP/s\n", name, m, fl / m / 1e9);
  printf("  partner role (the other wave of each SIMD), ms alone / together with the matrix role:\n");
  printf("    128 v_fma_f32 per iteration          alone %.3f   both %.3f\n", run<P_FMA, false, 0, BF16>(out), run<P_FMA, true, 0, BF16>(out));
  printf("    128 v_cvt_pk_bf16_f32 per iteration  alone %.3f   both %.3f\n", run<P_CVT, false, 0, BF16>(out), run<P_CVT, true, 0, BF16>(out));
  printf("    16 ds_read_b128 per iteration        alone %.3f   both %.3f\n", run<P_LDSR, false, 0, BF16>(out), run<P_LDSR, true, 0, BF16>(out));
  printf("    24 ds_write_b64 per iteration        alone %.3f   both %.3f\n", run<P_LDSW, false, 0, BF16>(out), run<P_LDSW, true, 0, BF16>(out));
  printf("  v_fma_f32 inside the matrix wave, INNER per MFMA, ms:  0: %.3f  2: %.3f  4: %.3f  6: %.3f  8: %.3f  12: %.3f\n", m,
         run<P_NONE, true, 2, BF16>(out), run<P_NONE, true, 4, BF16>(out), run<P_NONE, true, 6, BF16>(out), run<P_NONE, true, 8, BF16>(out),
         run<P_NONE, true, 12, BF16>(out));
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  report<true>(out, "v_mfma_f32_32x32x16_bf16 (8 passes)", 2.0 * 32 * 32 * 16);
  report<false>(out, "v_mfma_f32_32x32x2_f32 (16 passes)", 2.0 * 32 * 32 * 2);
  return 0;
}
