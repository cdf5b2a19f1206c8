"""Drop-in for the reference's data/util.py with transform_augment executed on the MI355X
(csrc/io_metrics.hip: sr3_images_u8_to_f32).  Same names: IMG_EXTENSIONS, is_image_file,
get_paths_from_images, transform_augment.  No torchvision.

data/util.py:76-83 of the reference: ToTensor on every PIL image, one RandomHorizontalFlip draw for the
stacked list when split == 'train', then img * (max - min) + min."""
import ctypes as C
import os

import numpy as np
import torch

from sr3_hip import lib as L

IMG_EXTENSIONS = ['.jpg', '.JPG', '.jpeg', '.JPEG', '.png', '.PNG', '.ppm', '.PPM', '.bmp', '.BMP']


def is_image_file(filename):
    return any(filename.endswith(ext) for ext in IMG_EXTENSIONS)


def get_paths_from_images(path):
    """Sorted list of every image file below `path` (data/util.py:16-25)."""
    assert os.path.isdir(path), '{:s} is not a valid directory'.format(path)
    images = [os.path.join(d, f) for d, _, fs in os.walk(path) for f in fs if is_image_file(f)]
    assert images, '{:s} has no valid image file'.format(path)
    return sorted(images)


def image_to_u8(img):
    """PIL image / array -> contiguous uint8 HWC torch tensor (what ToTensor reads; >3 channels are not expected
    after `.convert('RGB')`)."""
    a = np.array(img)              # a writable copy of the decoded pixels
    if a.dtype != np.uint8:
        raise TypeError('8-bit images expected, got %s' % a.dtype)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(a))


def u8_batch_to_f32(batch_u8, flip=None, min_max=(0, 1), device=None, out=None):
    """(n, H, W, C) uint8 (host or device) -> (n, C, H, W) fp32 on the device: / 255, optional per-image horizontal
    flip (uint8 / bool tensor of n flags), * (max - min) + min.  One kernel launch; the batch crosses PCIe as bytes."""
    if device is None:
        if not torch.cuda.is_available():
            raise L.Sr3Error('data.util runs on the MI355X engine: no GPU visible and there is no CPU fallback')
        device = torch.device('cuda', torch.cuda.current_device())
    if batch_u8.dtype != torch.uint8 or batch_u8.dim() != 4:
        raise TypeError('expected an (n, H, W, C) uint8 tensor')
    b = batch_u8.to(device, non_blocking=True).contiguous()
    n, H, W, Cc = b.shape
    f = None
    if flip is not None:
        f = torch.as_tensor(flip).to(torch.uint8).to(device, non_blocking=True).contiguous()
        if f.numel() != n:
            raise ValueError('one flip flag per image expected')
    if out is None:
        out = torch.empty((n, Cc, H, W), dtype=torch.float32, device=device)
    L.check(L.load().sr3_images_u8_to_f32(L.ptr(b), n, H, W, Cc, L.ptr(f), float(min_max[0]), float(min_max[1]), L.ptr(out),
                                          C.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
    return out


def transform_augment(img_list, split='val', min_max=(0, 1)):
    """Reference signature: list of PIL images -> list of (C, H, W) fp32 tensors (here: on the device)."""
    u8 = [image_to_u8(i) for i in img_list]
    flip = None
    if split == 'train':
        # RandomHorizontalFlip(p=0.5) on the stacked list: one draw for all of them (and torch.stack's
        # same-size requirement)
        if any(t.shape != u8[0].shape for t in u8):
            raise RuntimeError('stack expects each tensor to be equal size')
        flip = [bool(torch.rand(1).item() < 0.5)] * len(u8)
    if all(t.shape == u8[0].shape for t in u8):
        out = u8_batch_to_f32(torch.stack(u8, 0), flip, min_max)
        return list(torch.unbind(out, 0))
    return [u8_batch_to_f32(t[None], None, min_max)[0] for t in u8]
