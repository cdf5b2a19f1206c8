// Bicubic / bilinear resampling of uint8 images exactly as Pillow's ImagingResample does it (the arithmetic behind
// data/prepare_data.py:17-40: torchvision.transforms.functional.resize(PIL image, size, Image.BICUBIC) ->
// PIL.Image.resize): per output index a window [xmin, xmin + n) of normalised double-precision bicubic
// (a = -0.5) weights, scaled by max(scale, 1) for antialiased down-sampling, converted to 22-bit fixed point,
// a horizontal pass then a vertical pass over the uint8 intermediate, each rounding with +2^21 >> 22 and
// clamping to [0, 255].  Integer arithmetic throughout => bit-exact with Pillow (pinned against Pillow itself,
// which this image ships; tests/test_gpu_io.py, tests/test_oracle_io.py).
#include <stdint.h>

#include "sr3_common.h"
#include "../../include/sr3_io_mi355x.h"

namespace sr3 {
namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// separately rounded double ops (no fma contraction): the coefficient doubles must equal Pillow's
__device__ __forceinline__ double dmul(double a, double b) { double r = a * b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ double dadd(double a, double b) { double r = a + b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ double dsub(double a, double b) { double r = a - b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ double ddiv(double a, double b) { double r = a / b; asm volatile("" : "+v"(r)); return r; }

// Pillow Resample.c bicubic_filter, a = -0.5
__device__ double bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return dadd(dmul(dmul(dsub(dmul(dadd(a, 2.0), x), dadd(a, 3.0)), x), x), 1.0);
  if (x < 2.0) return dmul(dsub(dmul(dadd(dmul(dsub(x, 5.0), x), 8.0), x), 4.0), a);
  return 0.0;
}

// Pillow Resample.c bilinear_filter
__device__ double bilinear(double x) {
  if (x < 0.0) x = -x;
  return x < 1.0 ? dsub(1.0, x) : 0.0;
}

// precompute_coeffs + normalize_coeffs_8bpc for one axis: thread xx fills bounds[xx] = {xmin, n} and kk[xx][ksize]
template <bool CUBIC>
__global__ void k_resample_coeffs(int in_size, int out_size, int ksize, int* __restrict__ bounds, int* __restrict__ kk) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= out_size) return;
  const double scale = ddiv((double)in_size, (double)out_size);
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = dmul(CUBIC ? 2.0 : 1.0, filterscale);
  const double center = dmul(dadd((double)xx, 0.5), scale);          // in0 = 0
  const double ss = ddiv(1.0, filterscale);
  int xmin = (int)dadd(dsub(center, support), 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)dadd(dadd(center, support), 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  int* k = kk + (size_t)xx * ksize;
  double ww = 0.0;
  auto weight = [&](int x) {
    const double arg = dmul(dadd(dsub((double)(x + xmin), center), 0.5), ss);
    return CUBIC ? bicubic(arg) : bilinear(arg);
  };
  for (int x = 0; x < xmax; ++x) ww = dadd(ww, weight(x));
  for (int x = 0; x < ksize; ++x) {
    int q = 0;
    if (x < xmax) {
      double w = weight(x);
      if (ww != 0.0) w = ddiv(w, ww);
      const double s = dmul(w, (double)(1 << PRECISION_BITS));
      q = w < 0 ? (int)dadd(-0.5, s) : (int)dadd(0.5, s);
    }
    k[x] = q;
  }
  bounds[2 * xx] = xmin;
  bounds[2 * xx + 1] = xmax;
}

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= PRECISION_BITS;                      // arithmetic shift, as Pillow's lookup index
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one pass along `axis` (0: horizontal, 1: vertical) of n images (H, W, C) -> (OH, OW, C)
template <int AXIS>
__global__ __launch_bounds__(256) void k_resample_pass(const unsigned char* __restrict__ in, int H, int W, int C, int OH, int OW,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                       unsigned char* __restrict__ out) {
  const int img = blockIdx.y;
  const size_t in_per = (size_t)H * W * C, out_per = (size_t)OH * OW * C;
  const unsigned char* ib = in + img * in_per;
  unsigned char* ob = out + img * out_per;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < out_per; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int ox = (int)((i / C) % OW);
    const int oy = (int)(i / ((size_t)C * OW));
    const int o = AXIS == 0 ? ox : oy;
    const int lo = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = kk + (size_t)o * ksize;
    int acc = 1 << (PRECISION_BITS - 1);
    if (AXIS == 0) {
      const unsigned char* p = ib + ((size_t)oy * W + lo) * C + c;
      for (int x = 0; x < n; ++x) acc += (int)p[(size_t)x * C] * k[x];
    } else {
      const unsigned char* p = ib + ((size_t)lo * W + ox) * C + c;
      for (int y = 0; y < n; ++y) acc += (int)p[(size_t)y * W * C] * k[y];
    }
    ob[i] = clip8(acc);
  }
}

int ksize_for(int in_size, int out_size, bool cubic) {
  double scale = (double)in_size / out_size;
  if (scale < 1.0) scale = 1.0;
  return (int)ceil((cubic ? 2.0 : 1.0) * scale) * 2 + 1;
}

struct Layout { size_t bx, kx, by, ky, tmp, total; int ksx, ksy; };
Layout layout(int n, int H, int W, int C, int OH, int OW, bool cubic) {
  Layout l;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  l.ksx = ksize_for(W, OW, cubic); l.ksy = ksize_for(H, OH, cubic);
  size_t off = 0;
  l.bx = off; off += al((size_t)OW * 2 * sizeof(int));
  l.kx = off; off += al((size_t)OW * l.ksx * sizeof(int));
  l.by = off; off += al((size_t)OH * 2 * sizeof(int));
  l.ky = off; off += al((size_t)OH * l.ksy * sizeof(int));
  l.tmp = off; off += al((size_t)n * H * OW * C);            // after the horizontal pass: (H, OW, C)
  l.total = off;
  return l;
}

}  // namespace
}  // namespace sr3

using namespace sr3;

extern "C" {

size_t sr3_resize_scratch_bytes(int n_images, int H, int W, int C, int OH, int OW) {
  if (n_images <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return 0;
  return layout(n_images, H, W, C, OH, OW, true).total;      // the bicubic tables are the larger ones
}

int sr3_resize_u8(const uint8_t* in_hwc, int n_images, int H, int W, int C, int OH, int OW, int resample, void* scratch,
                  size_t scratch_bytes, uint8_t* out_hwc, void* stream) {
  if (resample != 2 && resample != 3) { set_error("resize: resample must be 2 (PIL BILINEAR) or 3 (PIL BICUBIC)"); return SR3_E_UNSUPPORTED; }
  const bool cubic = resample == 3;
  if (!in_hwc || !out_hwc || !scratch) { set_error("null argument"); return SR3_E_BADARG; }
  if (n_images <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) { set_error("resize: bad shape"); return SR3_E_BADARG; }
  const Layout l = layout(n_images, H, W, C, OH, OW, cubic);
  if (scratch_bytes < l.total) { set_error("resize: scratch too small (%zu < %zu)", scratch_bytes, l.total); return SR3_E_NOMEM; }
  if (((uintptr_t)scratch & 255)) { set_error("resize: scratch must be 256-byte aligned"); return SR3_E_ALIGN; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(scratch);
  int* bx = reinterpret_cast<int*>(ws + l.bx); int* kx = reinterpret_cast<int*>(ws + l.kx);
  int* by = reinterpret_cast<int*>(ws + l.by); int* ky = reinterpret_cast<int*>(ws + l.ky);
  unsigned char* tmp = reinterpret_cast<unsigned char*>(ws + l.tmp);
  const bool need_h = OW != W, need_v = OH != H;          // Pillow skips a pass whose size does not change
  auto blocks = [](size_t n) { size_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); };
  if (!need_h && !need_v) {
    SR3_HIP(hipMemcpyAsync(out_hwc, in_hwc, (size_t)n_images * H * W * C, hipMemcpyDeviceToDevice, st));
    return SR3_OK;
  }
  const unsigned char* src = in_hwc;
  if (need_h) {
    if (cubic) hipLaunchKernelGGL(k_resample_coeffs<true>, dim3((OW + 63) / 64), dim3(64), 0, st, W, OW, l.ksx, bx, kx);
    else hipLaunchKernelGGL(k_resample_coeffs<false>, dim3((OW + 63) / 64), dim3(64), 0, st, W, OW, l.ksx, bx, kx);
    unsigned char* dst = need_v ? tmp : out_hwc;
    hipLaunchKernelGGL(k_resample_pass<0>, dim3(blocks((size_t)H * OW * C), n_images), dim3(256), 0, st, src, H, W, C, H, OW, bx, kx,
                       l.ksx, dst);
    src = dst;
  }
  if (need_v) {
    if (cubic) hipLaunchKernelGGL(k_resample_coeffs<true>, dim3((OH + 63) / 64), dim3(64), 0, st, H, OH, l.ksy, by, ky);
    else hipLaunchKernelGGL(k_resample_coeffs<false>, dim3((OH + 63) / 64), dim3(64), 0, st, H, OH, l.ksy, by, ky);
    hipLaunchKernelGGL(k_resample_pass<1>, dim3(blocks((size_t)OH * OW * C), n_images), dim3(256), 0, st, src, H, OW, C, OH, OW, by, ky,
                       l.ksy, out_hwc);
  }
  SR3_LAUNCH_CHECK("k_resample_pass");
  return SR3_OK;
}

}  // extern "C"
