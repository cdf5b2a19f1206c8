"""Drop-in for the reference's core/metrics.py (tensor2img, save_img, calculate_psnr, calculate_ssim: same
names, arguments and return types) with the arithmetic on the MI355X: csrc/io_metrics.hip through
include/sr3_io_mi355x.h.  No cv2 / torchvision; there is no CPU fallback -- without a GPU these raise.

Engine extensions (not in the reference): `tensor2img_device` keeps the bytes on the GPU, and
`psnr_ssim_batch(sr, hr)` fuses the validation loop's per-image chain (sr.py:119-145, eval.py) for a whole
batch so that only two scalars per image cross PCIe."""
import ctypes as C
import math

import numpy as np
import torch

from sr3_hip import lib as L


def _device():
    if not torch.cuda.is_available():
        raise L.Sr3Error('core.metrics runs on the MI355X engine: no GPU visible and there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def tensor2img_device(tensor, out_type=np.uint8, min_max=(-1, 1)):
    """core/metrics.py:8-34 with the result left on the device: torch.uint8 (or float32 for other out_types),
    shape (H, W, C) / (H, W) exactly as the reference's numpy array."""
    dev = tensor.device if tensor.is_cuda else _device()
    t = tensor.squeeze().to(dev, torch.float32).contiguous()
    nd = t.dim()
    if nd == 4:
        n, Cc, H, W = t.shape
    elif nd == 3:
        n, (Cc, H, W) = 1, t.shape
    elif nd == 2:
        n, Cc, (H, W) = 1, 1, t.shape
    else:
        raise TypeError('Only support 4D, 3D and 2D tensor. But received with dimension: {:d}'.format(nd))
    lib = L.load()
    as_float = 0 if out_type == np.uint8 else 1
    oh, ow, oc = C.c_int(), C.c_int(), C.c_int()
    nrow = int(math.sqrt(n))
    args = (n, Cc, H, W, float(min_max[0]), float(min_max[1]), nrow, 2, as_float)
    L.check(lib.sr3_tensor2img(None, *args, None, C.byref(oh), C.byref(ow), C.byref(oc), None))
    out = torch.empty((oh.value, ow.value, oc.value), dtype=torch.float32 if as_float else torch.uint8, device=dev)
    L.check(lib.sr3_tensor2img(L.ptr(t), *args, L.ptr(out), None, None, None, _stream(dev)))
    if nd == 2:
        out = out[:, :, 0]
    return out


def tensor2img(tensor, out_type=np.uint8, min_max=(-1, 1)):
    '''
    Converts a torch Tensor into an image Numpy array
    Input: 4D(B,(3/1),H,W), 3D(C,H,W), or 2D(H,W), any range, RGB channel order
    Output: 3D(H,W,C) or 2D(H,W), [0,255], np.uint8 (default)
    '''
    return tensor2img_device(tensor, out_type, min_max).cpu().numpy().astype(out_type)


def save_img(img, img_path, mode='RGB'):
    """core/metrics.py:37-39 writes the RGB array (cv2.imwrite of the BGR-swapped copy); PIL writes it as is."""
    from PIL import Image
    from sr3_hip.dist import is_primary
    if not is_primary():     # data parallel: every rank holds the same validation images (ValWave); rank 0 writes them
        return
    Image.fromarray(np.ascontiguousarray(img)).save(img_path)


def _u8_pair(img1, img2, dev):
    if not tuple(img1.shape) == tuple(img2.shape):
        raise ValueError('Input images must have the same dimensions.')
    out = []
    for im in (img1, img2):
        if isinstance(im, np.ndarray):
            if im.dtype != np.uint8:
                raise TypeError('the engine metrics take uint8 images (what tensor2img produces); got %s' % im.dtype)
            im = torch.from_numpy(np.ascontiguousarray(im))
        elif im.dtype != torch.uint8:
            raise TypeError('the engine metrics take uint8 images; got %s' % im.dtype)
        out.append(im.to(dev).contiguous())
    return out


def _psnr_from_sse(sse, n):
    if sse == 0:
        return float('inf')
    mse = float(sse) / n                      # == np.mean((a - b)**2) in float64: exact integers
    return 20 * math.log10(255.0 / math.sqrt(mse))


def calculate_psnr(img1, img2):
    # img1 and img2 have range [0, 255] (uint8 arrays or device tensors)
    dev = img1.device if torch.is_tensor(img1) and img1.is_cuda else _device()
    a, b = _u8_pair(img1, img2, dev)
    sse = torch.zeros(1, dtype=torch.int64, device=dev)
    L.check(L.load().sr3_sse_u8(L.ptr(a), L.ptr(b), 1, a.numel(), L.ptr(sse), _stream(dev)))
    return _psnr_from_sse(int(sse.item()), a.numel())


def calculate_ssim(img1, img2):
    '''calculate SSIM
    the same outputs as MATLAB's
    img1, img2: [0, 255]
    '''
    dev = img1.device if torch.is_tensor(img1) and img1.is_cuda else _device()
    a, b = _u8_pair(img1, img2, dev)
    if a.dim() == 2:
        H, W, Cc = a.shape[0], a.shape[1], 1
    elif a.dim() == 3 and a.shape[2] in (1, 3):
        H, W, Cc = a.shape
    else:
        raise ValueError('Wrong input image dimensions.')
    lib = L.load()
    nb = int(lib.sr3_ssim_scratch_bytes(1, H, W, Cc))
    if nb == 0:
        raise ValueError('SSIM needs images of at least 11 x 11 pixels')
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    out = torch.empty(1, dtype=torch.float64, device=dev)
    L.check(lib.sr3_ssim_u8(L.ptr(a), L.ptr(b), 1, H, W, Cc, L.ptr(scratch), nb, L.ptr(out), _stream(dev)))
    return float(out.item())


def psnr_ssim_batch(sr, hr, min_max=(-1, 1)):
    """Engine extension: PSNR and SSIM of every image pair of two (B, C, H, W) fp32 tensors (device or host), each
    quantised exactly as tensor2img does.  Returns two Python lists of B floats."""
    dev = sr.device if sr.is_cuda else _device()
    if sr.dim() == 3:
        sr, hr = sr[None], hr[None]
    if sr.shape != hr.shape or sr.dim() != 4:
        raise ValueError('Input images must have the same dimensions.')
    sr = sr.to(dev, torch.float32).contiguous()
    hr = hr.to(dev, torch.float32).contiguous()
    B, Cc, H, W = sr.shape
    lib = L.load()
    nb = int(lib.sr3_eval_scratch_bytes(B, Cc, H, W))
    if H < 11 or W < 11 or nb == 0:
        raise ValueError('SSIM needs images of at least 11 x 11 pixels')
    scratch = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
    off = (-scratch.data_ptr()) % 256
    sse = torch.empty(B, dtype=torch.int64, device=dev)
    ssim = torch.empty(B, dtype=torch.float64, device=dev)
    L.check(lib.sr3_eval_psnr_ssim_f32(L.ptr(sr), L.ptr(hr), B, Cc, H, W, float(min_max[0]), float(min_max[1]),
                                       C.c_void_p(scratch.data_ptr() + off), nb, L.ptr(sse), L.ptr(ssim), _stream(dev)))
    n = Cc * H * W
    return [_psnr_from_sse(int(v), n) for v in sse.tolist()], ssim.tolist()
