/* sr3_io_mi355x.h -- C ABI of the steps either side of the SR3 hot path, exported by
 * the same libsr3_mi355x.so as include/sr3_mi355x.h (error codes, sr3_last_error() and the pointer / stream
 * conventions are the ones defined there: device pointers, asynchronous on `stream`, no allocation inside).
 *
 *   before the path: data/util.py:76-83 transform_augment (ToTensor, shared horizontal flip, range map), which
 *                    data/LRHR_dataset.py:92-99 applies to every sample of a batch
 *                    data/prepare_data.py:17-40 resize_multiple (the bicubic LR / HR / SR triplet a sample is made of)
 *   after the path : core/metrics.py:8-34 tensor2img, :43-50 calculate_psnr, :53-93 ssim / calculate_ssim, as used by
 *                    sr.py:119-145,188-196, infer.py:73-92 and eval.py on the tensors DDPM.get_current_visuals returns
 *
 * Byte and integer results are bit-exact with the reference's numpy/torch arithmetic; SSIM is evaluated in double with
 * a fixed summation order (the reference's cv2.filter2D is DFT based for an 11 x 11 kernel, so it carries ~1e-13 of its
 * own rounding noise).  The Python mirror is image-super-resolution-via-iterative-refinement_amd/core/metrics.py and
 * .../data/util.py. */
#ifndef SR3_IO_MI355X_H
#define SR3_IO_MI355X_H

#include <stddef.h>
#include <stdint.h>

#include "sr3_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tensor2img (core/metrics.py:8-34) for the 4-D / 3-D inputs: x is (n, C, H, W) fp32, any range.
 *   t = clamp(x, lo, hi); t = (t - lo) / (hi - lo)
 *   n == 1: out = t as (H, W, C)                               (the `squeeze()` -> 3-D branch, :25-27)
 *   n  > 1: out = make_grid(t, nrow, padding, pad_value 0) as (GH, GW, 3 if C == 1 else C)      (:19-24;
 *           the caller passes nrow = int(sqrt(n)) and padding = 2, torchvision's default)
 *   as_float == 0: out is uint8 = round_half_even(t * 255)      (:31-34);  else out is fp32 t (other `out_type`s)
 * out_h / out_w / out_c receive the output shape; with out_hwc == NULL the call is a pure size query. */
int sr3_tensor2img(const float* x_nchw, int n, int C, int H, int W, float lo, float hi, int nrow, int padding,
                   int as_float, void* out_hwc, int* out_h, int* out_w, int* out_c, void* stream);

/* calculate_psnr's reduction (core/metrics.py:43-50): sse_out[i] = sum over the bytes of image i of (a - b)^2, exact in
 * 64-bit integers.  np.mean((a - b)**2) in float64 is exactly sse / bytes_per_image, so the host finishes with the
 * reference's own expression 20 * log10(255 / sqrt(mse)) (inf when sse == 0). */
int sr3_sse_u8(const uint8_t* a, const uint8_t* b, int n_images, size_t bytes_per_image, unsigned long long* sse_out,
               void* stream);

/* ssim / calculate_ssim (core/metrics.py:53-93) of n_images HWC uint8 image pairs: 11 x 11 window =
 * outer(g, g), g = cv2.getGaussianKernel(11, 1.5); mu / sigma maps over the valid region [5:-5, 5:-5]; mean of the SSIM
 * map over all valid pixels and channels (calculate_ssim's 3-channel branch averages three identical whole-image
 * values, :85-88, i.e. the same number).  ssim_out[i] is a double on the device. */
size_t sr3_ssim_scratch_bytes(int n_images, int H, int W, int C);
int sr3_ssim_u8(const uint8_t* a_hwc, const uint8_t* b_hwc, int n_images, int H, int W, int C, void* scratch,
                size_t scratch_bytes, double* ssim_out, void* stream);

/* The validation loop's per-image chain (sr.py:119-145: tensor2img(SR), tensor2img(HR), calculate_psnr,
 * [calculate_ssim in eval.py / infer.py]) fused on the device for a batch of fp32 NCHW tensors in [lo, hi]: quantise
 * both to uint8 HWC exactly as tensor2img does, then the two reductions above.  Only n_images * 16 bytes return to the
 * host instead of two fp32 images per sample. */
size_t sr3_eval_scratch_bytes(int n_images, int C, int H, int W);
int sr3_eval_psnr_ssim_f32(const float* sr_nchw, const float* hr_nchw, int n_images, int C, int H, int W, float lo,
                           float hi, void* scratch, size_t scratch_bytes, unsigned long long* sse_out, double* ssim_out,
                           void* stream);

/* transform_augment (data/util.py:76-83) for a batch of decoded images: in is (n, H, W, C) uint8 (PIL RGB order), out
 * is (n, C, H, W) fp32:  v = in / 255 (ToTensor);  if flip[i]: reverse the W axis (RandomHorizontalFlip; the reference
 * draws ONE flag for the stacked [SR, HR] pair of a sample -- the caller repeats it);  out = v * (hi - lo) + lo with a
 * separately rounded multiply and add.  flip may be NULL (phase 'val'). */
int sr3_images_u8_to_f32(const uint8_t* in_hwc, int n_images, int H, int W, int C, const uint8_t* flip, float lo, float hi,
                         float* out_nchw, void* stream);

/* Data preparation (data/prepare_data.py:17-40 resize_and_convert / resize_multiple): PIL.Image.resize(size,
 * resample) of n_images uint8 (H, W, C) images to (OH, OW, C) for resample = 3 (Image.BICUBIC, the default of
 * prepare_data.py) or 2 (Image.BILINEAR, its `--resample bilinear`), bit-exact with Pillow's ImagingResample
 * (double-precision filter weights -- bicubic a = -0.5 -- over a support scaled by max(in/out, 1), normalised, 22-bit fixed
 * point, horizontal then vertical pass over a uint8 intermediate, each with round-to-nearest and clamp).  The
 * smaller-edge / centre-crop logic of torchvision's resize + center_crop lives in the Python mirror
 * (data/prepare_data.py of the package).  scratch: 256-byte aligned, sr3_resize_scratch_bytes(...) bytes. */
size_t sr3_resize_scratch_bytes(int n_images, int H, int W, int C, int OH, int OW);
int sr3_resize_u8(const uint8_t* in_hwc, int n_images, int H, int W, int C, int OH, int OW, int resample, void* scratch,
                  size_t scratch_bytes, uint8_t* out_hwc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SR3_IO_MI355X_H */
