#!/usr/bin/env python3
"""Generates tests/golden/io_metrics.npz by running the REFERENCE's own core/metrics.py and data/util.py
(imported from /root/reference, this container only) on seeded synthetic inputs.

cv2 and torchvision are not installed here, so the four third-party calls those two files make are stubbed
with the restatements of oracle/io_metrics_oracle.py (cv2.getGaussianKernel, cv2.filter2D, torchvision's
make_grid / ToTensor / RandomHorizontalFlip); everything else -- clamp / rescale / rounding in tensor2img, the
PSNR expression, the SSIM map and calculate_ssim's dispatch, the range map of transform_augment -- is the
reference's code executing unchanged.  tests/test_oracle_io.py additionally checks the stubs against scipy."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import io_metrics_oracle as O        # noqa: E402

REF = '/root/reference'
FLIP = {'on': False}


def install_stubs():
    cv2 = types.ModuleType('cv2')
    cv2.getGaussianKernel = O.gaussian_kernel

    def filter2D(img, ddepth, window):
        # full-size output like OpenCV (the reference crops [5:-5, 5:-5]); border values are never read
        r = window.shape[0] // 2
        out = np.zeros_like(np.asarray(img, dtype=np.float64))
        out[r:-r, r:-r] = O.filter2d_valid(img, window)
        return out
    cv2.filter2D = filter2D
    cv2.COLOR_RGB2BGR = 4
    cv2.cvtColor = lambda img, code: img[..., ::-1]
    cv2.imwrite = lambda path, img: True
    sys.modules['cv2'] = cv2

    tv = types.ModuleType('torchvision')
    tvu = types.ModuleType('torchvision.utils')
    tvu.make_grid = lambda tensor, nrow=8, normalize=False: torch.from_numpy(O.make_grid(tensor.numpy(), nrow=nrow))
    tvt = types.ModuleType('torchvision.transforms')

    class ToTensor(object):
        def __call__(self, pic):
            return torch.from_numpy(O.to_tensor(np.asarray(pic)))

    class RandomHorizontalFlip(object):
        def __call__(self, t):
            return t.flip(-1) if FLIP['on'] else t
    tvt.ToTensor = ToTensor
    tvt.RandomHorizontalFlip = RandomHorizontalFlip
    tv.utils = tvu
    tv.transforms = tvt
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.utils'] = tvu
    sys.modules['torchvision.transforms'] = tvt


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import core.metrics as M          # the reference's file
    import data.util as U             # the reference's file
    rng = np.random.RandomState(20240921)
    out = {}
    # ---- tensor2img ----
    x = (rng.randn(5, 3, 12, 14) * 0.8).astype(np.float32)
    x[0, 0, 0, :6] = [-1.5, 1.5, -1.0, 1.0, 0.0, 0.00392157]          # clamp / end points
    # exact .5 cases of t*255: t = (k + 0.5)/255 -> x = 2t - 1
    x[1, 1, 1, :8] = (2.0 * (np.arange(8) + 0.5) / 255.0 - 1.0).astype(np.float32)
    out['t2i/x'] = x
    out['t2i/single'] = M.tensor2img(torch.from_numpy(x[:1].copy()))
    out['t2i/chw'] = M.tensor2img(torch.from_numpy(x[2].copy()))
    out['t2i/grid5'] = M.tensor2img(torch.from_numpy(x.copy()))
    out['t2i/grid4_01'] = M.tensor2img(torch.from_numpy(x[:4].copy()), min_max=(0, 1))
    g1 = (rng.rand(3, 1, 9, 10).astype(np.float32) * 2 - 1)
    out['t2i/gray_x'] = g1
    out['t2i/gray_grid'] = M.tensor2img(torch.from_numpy(g1.copy()))
    out['t2i/gray_2d'] = M.tensor2img(torch.from_numpy(g1[:1].copy()))
    out['t2i/float_single'] = M.tensor2img(torch.from_numpy(x[:1].copy()), out_type=np.float32)
    # ---- psnr / ssim ----
    a = rng.randint(0, 256, size=(24, 20, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.randint(-12, 13, size=a.shape), 0, 255).astype(np.uint8)
    out['m/a'] = a
    out['m/b'] = b
    out['m/psnr'] = np.float64(M.calculate_psnr(a, b))
    out['m/psnr_same'] = np.float64(M.calculate_psnr(a, a))
    out['m/ssim3'] = np.float64(M.calculate_ssim(a, b))
    out['m/ssim1'] = np.float64(M.calculate_ssim(a[:, :, :1], b[:, :, :1]))
    out['m/ssim2d'] = np.float64(M.calculate_ssim(a[:, :, 0], b[:, :, 0]))
    # a smooth pair (realistic SSIM range)
    yy, xx = np.mgrid[0:40, 0:36]
    s = (127.5 + 100 * np.sin(yy / 5.0)[:, :, None] * np.cos(xx / 7.0)[:, :, None] * np.ones((1, 1, 3)))
    s8 = np.clip(s, 0, 255).astype(np.uint8)
    n8 = np.clip(s + rng.randn(*s.shape) * 9, 0, 255).astype(np.uint8)
    out['m/s'] = s8
    out['m/n'] = n8
    out['m/psnr_sn'] = np.float64(M.calculate_psnr(n8, s8))
    out['m/ssim_sn'] = np.float64(M.calculate_ssim(n8, s8))
    # ---- transform_augment ----
    imgs = [rng.randint(0, 256, size=(8, 9, 3)).astype(np.uint8) for _ in range(2)]
    out['tr/in0'], out['tr/in1'] = imgs
    r = U.transform_augment([im.copy() for im in imgs], split='val', min_max=(-1, 1))
    out['tr/val0'], out['tr/val1'] = r[0].numpy(), r[1].numpy()
    FLIP['on'] = True
    r = U.transform_augment([im.copy() for im in imgs], split='train', min_max=(-1, 1))
    out['tr/train_flip0'], out['tr/train_flip1'] = r[0].numpy(), r[1].numpy()
    FLIP['on'] = False
    r = U.transform_augment([im.copy() for im in imgs], split='train', min_max=(0, 1))
    out['tr/train_noflip01_0'] = r[0].numpy()
    dst = os.path.join(ROOT, 'tests', 'golden', 'io_metrics.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes;', len(out), 'arrays')


if __name__ == '__main__':
    main()
