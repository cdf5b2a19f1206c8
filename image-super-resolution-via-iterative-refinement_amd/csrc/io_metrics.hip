// The steps either side of the hot path, on the device so that a batch
// crosses PCIe as bytes, not as fp32, and quality numbers come back as a few scalars:
//   before: data/util.py:76-83 transform_augment = ToTensor (u8 HWC -> f32 CHW / 255), one shared
//           RandomHorizontalFlip draw, then x * (max - min) + min
//   after : core/metrics.py:8-34 tensor2img (clamp, rescale, make_grid for batches, *255, round, u8),
//           :43-50 calculate_psnr, :53-93 ssim / calculate_ssim (11x11 Gaussian sigma 1.5, "valid" crop)
// Integer / byte results are bit-exact with the reference's numpy arithmetic (separately rounded fp32
// ops, round-half-even); SSIM is double precision with a fixed summation order.
#include <stdint.h>

#include "sr3_common.h"
#include "../../include/sr3_io_mi355x.h"

namespace sr3 {
namespace {

// separately rounded fp32 ops (hipcc contracts a*b+c into fma by default; numpy / torch do not)
__device__ __forceinline__ float mul_r(float a, float b) { float r = a * b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ float add_r(float a, float b) { float r = a + b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ float sub_r(float a, float b) { float r = a - b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ float div_r(float a, float b) { float r = a / b; asm volatile("" : "+v"(r)); return r; }

// metrics.py:14-16: clamp_(min, max); (t - min) / (max - min)
__device__ __forceinline__ float unit_range(float x, float lo, float hi) {
  const float t = fminf(fmaxf(x, lo), hi);
  return div_r(sub_r(t, lo), sub_r(hi, lo));
}
// metrics.py:31-34: (img * 255.0).round().astype(uint8) -- numpy rounds half to even
__device__ __forceinline__ unsigned char quant_u8(float u) { return (unsigned char)rintf(mul_r(u, 255.0f)); }

struct GridGeom {
  int n, C, H, W;          // source batch (n, C, H, W)
  int OC;                  // output channels (make_grid repeats a single channel 3x)
  int xmaps, pad;          // images per grid row, padding (0: no grid, n == 1)
  int GH, GW;              // output height / width
};

// out[gy][gx][c] (HWC).  AS_U8: quantised bytes; else the [0,1] floats (tensor2img with another out_type)
template <bool AS_U8>
__global__ __launch_bounds__(256) void k_tensor2img(const float* __restrict__ x, GridGeom g, float lo, float hi, void* __restrict__ out) {
  const size_t total = (size_t)g.GH * g.GW * g.OC;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % g.OC);
    const size_t pix = i / g.OC;
    const int gx = (int)(pix % g.GW), gy = (int)(pix / g.GW);
    float u = 0.f;                       // make_grid pad_value 0 (applied to the already rescaled tensor)
    int k = 0, y = gy, xx = gx;
    bool inside = true;
    if (g.pad > 0) {
      const int ch = g.H + g.pad, cw = g.W + g.pad;
      const int ry = gy - g.pad, rx = gx - g.pad;
      inside = ry >= 0 && rx >= 0;
      const int cy = inside ? ry / ch : 0, cx = inside ? rx / cw : 0;
      y = ry - cy * ch; xx = rx - cx * cw;
      k = cy * g.xmaps + cx;
      inside = inside && y < g.H && xx < g.W && cx < g.xmaps && k < g.n;
    }
    if (inside) {
      const int cs = g.C == 1 ? 0 : c;
      u = unit_range(x[(((size_t)k * g.C + cs) * g.H + y) * g.W + xx], lo, hi);
    }
    if (AS_U8) static_cast<unsigned char*>(out)[i] = quant_u8(u);
    else static_cast<float*>(out)[i] = u;
  }
}

// per-image NCHW f32 -> HWC u8 (the n == 1 path of tensor2img applied to every image of a batch)
__global__ __launch_bounds__(256) void k_batch_to_u8(const float* __restrict__ x, int C, int H, int W, float lo, float hi,
                                                     unsigned char* __restrict__ out) {
  const int b = blockIdx.y;
  const size_t per = (size_t)C * H * W;
  const float* xb = x + b * per;
  unsigned char* ob = out + b * per;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t pix = i / C;
    ob[i] = quant_u8(unit_range(xb[(size_t)c * H * W + pix], lo, hi));
  }
}

// sum of squared byte differences per image, exact in u64 (metrics.py:45-47: np.mean((a - b)**2) in float64
// is the same integer divided by n)
__global__ __launch_bounds__(256) void k_sse_u8(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b, size_t per,
                                                unsigned long long* __restrict__ sse) {
  const int img = blockIdx.y;
  const unsigned char* pa = a + img * per;
  const unsigned char* pb = b + img * per;
  unsigned long long acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)pa[i] - (int)pb[i];
    acc += (unsigned)(d * d);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(sse + img, acc);
}

struct Gauss11 { double k[11]; };

// SSIM map of one HWC u8 image pair over the "valid" region (metrics.py:61-72), summed per block.
// Block = 16x16 output pixels of one channel plane; LDS holds the 26x26 input patches as doubles.
constexpr int ST = 16, SP = ST + 10;
__global__ __launch_bounds__(256) void k_ssim_u8(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b, int H, int W, int C,
                                                 Gauss11 gk, double* __restrict__ partial) {
  __shared__ double sa[SP][SP + 1], sb[SP][SP + 1];
  __shared__ double red[4];
  const int VH = H - 10, VW = W - 10;
  const int tiles_x = (VW + ST - 1) / ST, tiles_y = (VH + ST - 1) / ST;
  int bid = blockIdx.x;
  const int tx_i = bid % tiles_x; bid /= tiles_x;
  const int ty_i = bid % tiles_y; bid /= tiles_y;
  const int c = bid;
  const int img = blockIdx.y;
  const size_t per = (size_t)H * W * C;
  const unsigned char* pa = a + img * per;
  const unsigned char* pb = b + img * per;
  const int y0 = ty_i * ST, x0 = tx_i * ST;      // top-left of the patch in image coordinates
  for (int i = threadIdx.x; i < SP * SP; i += 256) {
    const int py = i / SP, px = i - py * SP;
    const int y = y0 + py, x = x0 + px;
    double va = 0.0, vb = 0.0;
    if (y < H && x < W) {
      const size_t o = ((size_t)y * W + x) * C + c;
      va = (double)pa[o]; vb = (double)pb[o];
    }
    sa[py][px] = va; sb[py][px] = vb;
  }
  __syncthreads();
  const int ly = threadIdx.x / ST, lx = threadIdx.x % ST;
  double val = 0.0;
  if (y0 + ly < VH && x0 + lx < VW) {
    double m1 = 0.0, m2 = 0.0, s11 = 0.0, s22 = 0.0, s12 = 0.0;
    for (int i = 0; i < 11; ++i) {
#pragma unroll
      for (int j = 0; j < 11; ++j) {
        const double w = gk.k[i] * gk.k[j];          // window = outer(kernel, kernel)  (metrics.py:59-60)
        const double p = sa[ly + i][lx + j], q = sb[ly + i][lx + j];
        m1 += w * p; m2 += w * q; s11 += w * (p * p); s22 += w * (q * q); s12 += w * (p * q);
      }
    }
    const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
    const double m1s = m1 * m1, m2s = m2 * m2, m12 = m1 * m2;
    const double v1 = s11 - m1s, v2 = s22 - m2s, cov = s12 - m12;
    val = ((2 * m12 + C1) * (2 * cov + C2)) / ((m1s + m2s + C1) * (v1 + v2 + C2));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) val += __shfl_xor(val, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = val;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)img * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// fixed-order sum of the block partials of each image -> mean over the valid map
__global__ __launch_bounds__(256) void k_ssim_finish(const double* __restrict__ partial, int nblocks, double inv_count, double* __restrict__ out) {
  __shared__ double red[256];
  const int img = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[(size_t)img * nblocks + i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[img] = red[0] * inv_count;
}

// transform_augment (data/util.py:76-83): ToTensor (u8 -> f32 / 255, HWC -> CHW), optional horizontal flip,
// then x * (max - min) + min with separately rounded multiply and add
__global__ __launch_bounds__(256) void k_u8_to_f32(const unsigned char* __restrict__ in, int C, int H, int W, const unsigned char* __restrict__ flip,
                                                   float lo, float hi, float* __restrict__ out) {
  const int b = blockIdx.y;
  const size_t per = (size_t)C * H * W;
  const unsigned char* ib = in + b * per;
  float* ob = out + b * per;
  const bool fl = flip && flip[b];
  const float span = sub_r(hi, lo);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
    // i indexes the CHW output (coalesced stores); the byte gather is served by L2
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int c = (int)(i / ((size_t)W * H));
    const int xs = fl ? W - 1 - x : x;
    const float v = div_r((float)ib[((size_t)y * W + xs) * C + c], 255.0f);
    ob[i] = add_r(mul_r(v, span), lo);
  }
}

inline int blocks_for(size_t n, int cap = 256 * 8) {
  const size_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

Gauss11 gaussian11() {
  // cv2.getGaussianKernel(11, 1.5): exp(-(i - 5)^2 / (2 sigma^2)), normalised to sum 1, in double
  Gauss11 g;
  double sum = 0.0;
  for (int i = 0; i < 11; ++i) { const double d = i - 5.0; g.k[i] = exp(-(d * d) / (2.0 * 1.5 * 1.5)); sum += g.k[i]; }
  for (int i = 0; i < 11; ++i) g.k[i] /= sum;
  return g;
}

}  // namespace
}  // namespace sr3

using namespace sr3;

extern "C" {

int sr3_tensor2img(const float* x_nchw, int n, int C, int H, int W, float lo, float hi, int nrow, int padding,
                   int as_float, void* out_hwc, int* out_h, int* out_w, int* out_c, void* stream) {
  if (n <= 0 || C <= 0 || H <= 0 || W <= 0 || !(hi > lo)) { set_error("tensor2img: bad shape or range"); return SR3_E_BADARG; }
  GridGeom g;
  g.n = n; g.C = C; g.H = H; g.W = W;
  if (n == 1) {
    g.OC = C; g.xmaps = 1; g.pad = 0; g.GH = H; g.GW = W;
  } else {
    // torchvision.utils.make_grid(tensor, nrow, padding=2, pad_value=0): single-channel images become 3-channel
    if (nrow <= 0 || padding < 0) { set_error("tensor2img: bad grid arguments"); return SR3_E_BADARG; }
    g.OC = C == 1 ? 3 : C;
    g.xmaps = nrow < n ? nrow : n;
    const int ymaps = (n + g.xmaps - 1) / g.xmaps;
    g.pad = padding;
    g.GH = (H + padding) * ymaps + padding;
    g.GW = (W + padding) * g.xmaps + padding;
    if (padding == 0) { set_error("tensor2img: grids need padding > 0 in this build"); return SR3_E_UNSUPPORTED; }
  }
  if (out_h) *out_h = g.GH;
  if (out_w) *out_w = g.GW;
  if (out_c) *out_c = g.OC;
  if (!out_hwc) return SR3_OK;                 // size query
  if (!x_nchw) { set_error("null argument"); return SR3_E_BADARG; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t total = (size_t)g.GH * g.GW * g.OC;
  if (as_float) hipLaunchKernelGGL(k_tensor2img<false>, dim3(blocks_for(total)), dim3(256), 0, st, x_nchw, g, lo, hi, out_hwc);
  else hipLaunchKernelGGL(k_tensor2img<true>, dim3(blocks_for(total)), dim3(256), 0, st, x_nchw, g, lo, hi, out_hwc);
  SR3_LAUNCH_CHECK("k_tensor2img");
  return SR3_OK;
}

int sr3_sse_u8(const uint8_t* a, const uint8_t* b, int n_images, size_t bytes_per_image, unsigned long long* sse_out, void* stream) {
  if (!a || !b || !sse_out || n_images <= 0) { set_error("null argument"); return SR3_E_BADARG; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  SR3_HIP(hipMemsetAsync(sse_out, 0, sizeof(unsigned long long) * n_images, st));
  if (bytes_per_image == 0) return SR3_OK;
  hipLaunchKernelGGL(k_sse_u8, dim3(blocks_for(bytes_per_image, 256), n_images), dim3(256), 0, st, a, b, bytes_per_image, sse_out);
  SR3_LAUNCH_CHECK("k_sse_u8");
  return SR3_OK;
}

size_t sr3_ssim_scratch_bytes(int n_images, int H, int W, int C) {
  if (H < 11 || W < 11 || C <= 0 || n_images <= 0) return 0;
  const size_t blocks = (size_t)((W - 10 + ST - 1) / ST) * ((H - 10 + ST - 1) / ST) * C;
  return blocks * n_images * sizeof(double);
}

int sr3_ssim_u8(const uint8_t* a_hwc, const uint8_t* b_hwc, int n_images, int H, int W, int C, void* scratch,
                size_t scratch_bytes, double* ssim_out, void* stream) {
  if (!a_hwc || !b_hwc || !ssim_out || !scratch) { set_error("null argument"); return SR3_E_BADARG; }
  if (H < 11 || W < 11 || C <= 0 || n_images <= 0) { set_error("ssim: images must be at least 11 x 11"); return SR3_E_BADARG; }
  const size_t need = sr3_ssim_scratch_bytes(n_images, H, W, C);
  if (scratch_bytes < need) { set_error("ssim: scratch too small (%zu < %zu)", scratch_bytes, need); return SR3_E_NOMEM; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int blocks = (int)(need / sizeof(double) / n_images);
  static const Gauss11 gk = gaussian11();
  hipLaunchKernelGGL(k_ssim_u8, dim3(blocks, n_images), dim3(256), 0, st, a_hwc, b_hwc, H, W, C, gk, static_cast<double*>(scratch));
  SR3_LAUNCH_CHECK("k_ssim_u8");
  const double inv = 1.0 / ((double)(H - 10) * (W - 10) * C);
  hipLaunchKernelGGL(k_ssim_finish, dim3(n_images), dim3(256), 0, st, static_cast<const double*>(scratch), blocks, inv, ssim_out);
  SR3_LAUNCH_CHECK("k_ssim_finish");
  return SR3_OK;
}

size_t sr3_eval_scratch_bytes(int n_images, int C, int H, int W) {
  if (n_images <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  const size_t img = ((size_t)n_images * C * H * W + 255) & ~(size_t)255;
  return 2 * img + sr3_ssim_scratch_bytes(n_images, H, W, C);
}

int sr3_eval_psnr_ssim_f32(const float* sr_nchw, const float* hr_nchw, int n_images, int C, int H, int W, float lo, float hi,
                           void* scratch, size_t scratch_bytes, unsigned long long* sse_out, double* ssim_out, void* stream) {
  if (!sr_nchw || !hr_nchw || !scratch || !sse_out || !ssim_out) { set_error("null argument"); return SR3_E_BADARG; }
  if (!(hi > lo)) { set_error("eval: bad range"); return SR3_E_BADARG; }
  const size_t need = sr3_eval_scratch_bytes(n_images, C, H, W);
  if (need == 0 || scratch_bytes < need) { set_error("eval: scratch too small (%zu < %zu)", scratch_bytes, need); return SR3_E_NOMEM; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t per = (size_t)C * H * W;
  const size_t img = ((size_t)n_images * per + 255) & ~(size_t)255;
  unsigned char* ua = static_cast<unsigned char*>(scratch);
  unsigned char* ub = ua + img;
  hipLaunchKernelGGL(k_batch_to_u8, dim3(blocks_for(per, 256), n_images), dim3(256), 0, st, sr_nchw, C, H, W, lo, hi, ua);
  hipLaunchKernelGGL(k_batch_to_u8, dim3(blocks_for(per, 256), n_images), dim3(256), 0, st, hr_nchw, C, H, W, lo, hi, ub);
  SR3_LAUNCH_CHECK("k_batch_to_u8");
  int rc = sr3_sse_u8(ua, ub, n_images, per, sse_out, stream);
  if (rc) return rc;
  return sr3_ssim_u8(ua, ub, n_images, H, W, C, ub + img, scratch_bytes - 2 * img, ssim_out, stream);
}

int sr3_images_u8_to_f32(const uint8_t* in_hwc, int n_images, int H, int W, int C, const uint8_t* flip, float lo, float hi,
                         float* out_nchw, void* stream) {
  if (!in_hwc || !out_nchw || n_images <= 0 || H <= 0 || W <= 0 || C <= 0) { set_error("images_u8_to_f32: bad argument"); return SR3_E_BADARG; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t per = (size_t)C * H * W;
  hipLaunchKernelGGL(k_u8_to_f32, dim3(blocks_for(per, 1024), n_images), dim3(256), 0, st, in_hwc, C, H, W, flip, lo, hi, out_nchw);
  SR3_LAUNCH_CHECK("k_u8_to_f32");
  return SR3_OK;
}

}  // extern "C"
