"""CPU restatement of the steps either side of the hot path.  TEST INFRASTRUCTURE ONLY:
imported by tests/, never by the product path (which calls the HIP kernels of csrc/io_metrics.hip through
include/sr3_io_mi355x.h and has no CPU fallback).

Follows, line by line, the reference's
  core/metrics.py:8-34   tensor2img        :43-50 calculate_psnr      :53-93 ssim / calculate_ssim
  data/util.py:76-83     transform_augment (ToTensor, shared RandomHorizontalFlip, range map)
Third-party pieces the reference calls but this image does not ship (cv2, torchvision) are restated from their
documented behaviour, each marked below.  Pinning: tests/golden/io_metrics.npz is produced by oracle/make_golden_io.py
by running the reference's OWN core/metrics.py and data/util.py code with those third-party calls stubbed by the
restatements here -- so the reference's control flow / arithmetic around them is pinned, the stubs themselves are
checked against scipy.ndimage (an independent implementation) in tests/test_oracle_io.py."""
import math

import numpy as np


# ---- torchvision.utils.make_grid (restated: torchvision is not installed) -------------------------------------
def make_grid(t, nrow=8, padding=2, pad_value=0.0):
    """t: (n, C, H, W) float array.  torchvision 0.x semantics: single-channel -> 3 channels; n == 1 -> the image;
    xmaps = min(nrow, n), ymaps = ceil(n / xmaps); cells of (H + padding, W + padding) on a pad_value canvas."""
    t = np.asarray(t)
    if t.shape[1] == 1:
        t = np.concatenate([t, t, t], axis=1)
    n, C, H, W = t.shape
    if n == 1:
        return t[0]
    xmaps = min(nrow, n)
    ymaps = int(math.ceil(float(n) / xmaps))
    height, width = H + padding, W + padding
    grid = np.full((C, height * ymaps + padding, width * xmaps + padding), pad_value, dtype=t.dtype)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[:, y * height + padding:y * height + padding + H, x * width + padding:x * width + padding + W] = t[k]
            k += 1
    return grid


# ---- core/metrics.py:8-34 ---------------------------------------------------------------------------------------
def tensor2img(tensor, out_type=np.uint8, min_max=(-1, 1)):
    t = np.asarray(tensor, dtype=np.float32)
    t = np.squeeze(t)                                                  # :14 tensor.squeeze()
    lo, hi = np.float32(min_max[0]), np.float32(min_max[1])
    t = np.clip(t, lo, hi)                                             # :14 clamp_
    t = (t - lo) / (hi - lo)                                           # :15-16, fp32
    if t.ndim == 4:
        n_img = len(t)
        img = make_grid(t, nrow=int(math.sqrt(n_img)))                 # :19-21
        img = np.transpose(img, (1, 2, 0))
    elif t.ndim == 3:
        img = np.transpose(t, (1, 2, 0))                               # :25-26
    elif t.ndim == 2:
        img = t
    else:
        raise TypeError('Only support 4D, 3D and 2D tensor. But received with dimension: {:d}'.format(t.ndim))
    if out_type == np.uint8:
        img = (img * np.float32(255.0)).round()                        # :32 numpy rounds half to even
    return img.astype(out_type)


# ---- core/metrics.py:43-50 -------------------------------------------------------------------------------------
def calculate_psnr(img1, img2):
    a = img1.astype(np.float64)
    b = img2.astype(np.float64)
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float('inf')
    return 20 * math.log10(255.0 / math.sqrt(mse))


# ---- cv2.getGaussianKernel / cv2.filter2D (restated: OpenCV is not installed) -----------------------------------
def gaussian_kernel(ksize=11, sigma=1.5):
    """cv2.getGaussianKernel(ksize, sigma) for sigma > 0: exp(-(i - (ksize-1)/2)^2 / (2 sigma^2)), sum-normalised;
    returned as a (ksize, 1) float64 column like OpenCV."""
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    return (k / k.sum()).reshape(ksize, 1)


def filter2d_valid(img, window):
    """cv2.filter2D(img, -1, window)[r:-r, r:-r]: correlation (no kernel flip), anchor at the centre; the border
    mode never matters because the caller crops the valid region.  img: (H, W) or (H, W, C) float64."""
    kh, kw = window.shape
    img = np.asarray(img, dtype=np.float64)
    H, W = img.shape[:2]
    out = np.zeros((H - kh + 1, W - kw + 1) + img.shape[2:], dtype=np.float64)
    for i in range(kh):
        for j in range(kw):
            out += window[i, j] * img[i:i + H - kh + 1, j:j + W - kw + 1]
    return out


# ---- core/metrics.py:53-93 ---------------------------------------------------------------------------------------
def ssim(img1, img2):
    C1 = (0.01 * 255) ** 2
    C2 = (0.03 * 255) ** 2
    a = img1.astype(np.float64)
    b = img2.astype(np.float64)
    kernel = gaussian_kernel(11, 1.5)
    window = np.outer(kernel, kernel.transpose())
    mu1 = filter2d_valid(a, window)
    mu2 = filter2d_valid(b, window)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    sigma1_sq = filter2d_valid(a ** 2, window) - mu1_sq
    sigma2_sq = filter2d_valid(b ** 2, window) - mu2_sq
    sigma12 = filter2d_valid(a * b, window) - mu1_mu2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean()


def calculate_ssim(img1, img2):
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    if img1.ndim == 2:
        return ssim(img1, img2)
    elif img1.ndim == 3:
        if img1.shape[2] == 3:
            return np.array([ssim(img1, img2) for _ in range(3)]).mean()     # :85-88 three identical whole-image values
        elif img1.shape[2] == 1:
            return ssim(np.squeeze(img1), np.squeeze(img2))
    raise ValueError('Wrong input image dimensions.')


# ---- data/util.py:76-83 (torchvision.transforms.ToTensor / RandomHorizontalFlip restated) -------------------------
def to_tensor(img_u8_hwc):
    """ToTensor on a PIL RGB image / uint8 HWC array: CHW float32 = byte / 255 (fp32 division)."""
    a = np.asarray(img_u8_hwc)
    if a.ndim == 2:
        a = a[:, :, None]
    return np.transpose(a, (2, 0, 1)).astype(np.float32) / np.float32(255)


def transform_augment(img_list, split='val', min_max=(0, 1), flip=False):
    """`flip` stands for the ONE Bernoulli(0.5) draw RandomHorizontalFlip makes for the stacked list (split 'train')."""
    imgs = [to_tensor(i) for i in img_list]
    if split == 'train':
        st = np.stack(imgs, 0)
        if flip:
            st = st[..., ::-1]
        imgs = list(st)
    span = min_max[1] - min_max[0]
    return [(i * np.float32(span) + np.float32(min_max[0])).astype(np.float32) for i in imgs]


# ---- data/prepare_data.py:17-40 -> PIL.Image.resize: Pillow's ImagingResample for 8-bit images, restated ------------
# (Pillow itself is installed in this image, so tests/test_oracle_io.py pins this restatement against the real
# `Image.resize` for many size pairs; libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
# ImagingResampleHorizontal_8bpc / Vertical_8bpc.)
PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def resample_coeffs(in_size, out_size, cubic=True):
    filt, fsupport = (_bicubic, 2.0) if cubic else (_bilinear, 1.0)
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img, out_size, axis, cubic):
    img = np.moveaxis(np.asarray(img, dtype=np.int64), axis, 0)
    bounds, kk = resample_coeffs(img.shape[0], out_size, cubic)
    out = np.zeros((out_size,) + img.shape[1:], dtype=np.int64)
    for xx in range(out_size):
        lo, n = bounds[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += img[lo + x] * kk[xx, x]
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def pil_resize(img_u8_hwc, out_hw, cubic=True):
    """Image.resize((OW, OH), BICUBIC | BILINEAR) of an (H, W, C) uint8 array: horizontal pass, then vertical."""
    a = np.asarray(img_u8_hwc)
    if a.shape[1] != out_hw[1]:
        a = _resample_axis(a, out_hw[1], 1, cubic)
    if a.shape[0] != out_hw[0]:
        a = _resample_axis(a, out_hw[0], 0, cubic)
    return a
