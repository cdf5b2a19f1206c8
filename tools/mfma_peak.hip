// Micro-benchmark: what does v_mfma_f32_32x32x2_f32 sustain on this chip with the halo kernel's occupancy
// (2 waves per SIMD, 4 independent accumulators per wave), with and without a ds_read_b128 stream beside it?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDS_READS>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  __shared__ f32x4 buf[1024];
  const int lane = threadIdx.x;
  buf[lane] = f32x4{1.f, 2.f, 3.f, 4.f};
  buf[lane + 256] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x4 a = buf[lane], b = buf[lane + 256];
  for (int it = 0; it < iters; ++it) {
    if (LDS_READS) {
#pragma unroll
      for (int q = 0; q < LDS_READS; ++q) {
        const f32x4 t = buf[(lane + 17 * q + it) & 511];
        a += t;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[(q + i) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int L>
void run(const char* name, float* out) {
  const int iters = 4096, blocks = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, out, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)blocks * 4 * iters * 16 * 4096.0;
  printf("%-34s %.3f ms  %.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 256 * 4);
  run<0>("mfma only", out);
  run<4>("mfma + 4 ds_read_b128 per 16 mfma", out);
  run<8>("mfma + 8 ds_read_b128 per 16 mfma", out);
  return 0;
}
