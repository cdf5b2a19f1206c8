"""Drop-in for the reference's data/prepare_data.py (resize_and_convert, image_convert_bytes, resize_multiple,
resize_worker, prepare and the same command line) with the resampling on the MI355X: csrc/resize.hip reproduces
Pillow's BICUBIC / BILINEAR `Image.resize` bit for bit, so the LR / HR / SR triplets are byte-identical to the
reference's.  torchvision's `resize` (smaller edge -> size, aspect kept) and `center_crop` are restated here
(torchvision is not needed).  Engine extension: `resize_batch` resizes a whole uint8 batch in one call; `prepare`
runs in one process (the GPU does the arithmetic; `n_worker` is accepted and ignored)."""
import argparse
import ctypes as C
import os
from io import BytesIO
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from sr3_hip import lib as L

_RESAMPLE = {Image.BILINEAR: 2, Image.BICUBIC: 3, 2: 2, 3: 3}


def _device():
    if not torch.cuda.is_available():
        raise L.Sr3Error('data.prepare_data runs on the MI355X engine: no GPU visible and there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def resize_batch(batch_u8, out_hw, resample=Image.BICUBIC):
    """(n, H, W, C) uint8 tensor (host or device) -> (n, OH, OW, C) uint8 on the device, = PIL resize per image."""
    if resample not in _RESAMPLE:
        raise NotImplementedError('resample %r: only Image.BICUBIC and Image.BILINEAR are built' % (resample,))
    dev = batch_u8.device if batch_u8.is_cuda else _device()
    b = batch_u8.to(dev).contiguous()
    n, H, W, Cc = b.shape
    OH, OW = int(out_hw[0]), int(out_hw[1])
    lib = L.load()
    nb = int(lib.sr3_resize_scratch_bytes(n, H, W, Cc, OH, OW))
    scratch = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
    off = (-scratch.data_ptr()) % 256
    out = torch.empty((n, OH, OW, Cc), dtype=torch.uint8, device=dev)
    L.check(lib.sr3_resize_u8(L.ptr(b), n, H, W, Cc, OH, OW, _RESAMPLE[resample], C.c_void_p(scratch.data_ptr() + off), nb,
                              L.ptr(out), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out


def _resized_hw(w, h, size):
    """torchvision.transforms.functional.resize with an int size: the smaller edge becomes `size`."""
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def _center_crop_box(w, h, size):
    """torchvision center_crop: (left, top) of the size x size window."""
    if w < size or h < size:
        raise NotImplementedError('center_crop padding of images smaller than the crop is not built')
    return int(round((w - size) / 2.0)), int(round((h - size) / 2.0))


def resize_and_convert(img, size, resample):
    """data/prepare_data.py:17-21: PIL image in, PIL image out."""
    if img.size[0] != size:
        w, h = img.size
        oh, ow = _resized_hw(w, h, size)
        a = torch.from_numpy(np.array(img))
        if a.dim() == 2:
            a = a[:, :, None]
        r = resize_batch(a[None], (oh, ow), resample)[0] if (oh, ow) != (h, w) else a
        left, top = _center_crop_box(ow, oh, size)
        r = r[top:top + size, left:left + size].cpu().numpy()
        img = Image.fromarray(r[:, :, 0] if r.shape[2] == 1 else np.ascontiguousarray(r))
    return img


def image_convert_bytes(img):
    buffer = BytesIO()
    img.save(buffer, format='png')
    return buffer.getvalue()


def resize_multiple(img, sizes=(16, 128), resample=Image.BICUBIC, lmdb_save=False):
    lr_img = resize_and_convert(img, sizes[0], resample)
    hr_img = resize_and_convert(img, sizes[1], resample)
    sr_img = resize_and_convert(lr_img, sizes[1], resample)
    if lmdb_save:
        lr_img, hr_img, sr_img = image_convert_bytes(lr_img), image_convert_bytes(hr_img), image_convert_bytes(sr_img)
    return [lr_img, hr_img, sr_img]


def resize_worker(img_file, sizes, resample, lmdb_save=False):
    img = Image.open(img_file).convert('RGB')
    out = resize_multiple(img, sizes=sizes, resample=resample, lmdb_save=lmdb_save)
    return Path(img_file).name.split('.')[0], out


def prepare(img_path, out_path, n_worker=1, sizes=(16, 128), resample=Image.BICUBIC, lmdb_save=False):
    files = sorted(p for p in Path('{}'.format(img_path)).glob('**/*') if p.is_file())
    env = None
    if not lmdb_save:
        for sub in ('lr_{}'.format(sizes[0]), 'hr_{}'.format(sizes[1]), 'sr_{}_{}'.format(sizes[0], sizes[1])):
            os.makedirs(os.path.join(out_path, sub), exist_ok=True)
    else:
        try:
            import lmdb
        except ImportError as e:
            raise ImportError('--lmdb needs the lmdb module (not installed here)') from e
        env = lmdb.open(out_path, map_size=1024 ** 4, readahead=False)
    total = 0
    for file in files:
        i, (lr_img, hr_img, sr_img) = resize_worker(file, sizes, resample, lmdb_save)
        key = i.zfill(5)
        if not lmdb_save:
            lr_img.save('{}/lr_{}/{}.png'.format(out_path, sizes[0], key))
            hr_img.save('{}/hr_{}/{}.png'.format(out_path, sizes[1], key))
            sr_img.save('{}/sr_{}_{}/{}.png'.format(out_path, sizes[0], sizes[1], key))
        else:
            with env.begin(write=True) as txn:
                txn.put('lr_{}_{}'.format(sizes[0], key).encode('utf-8'), lr_img)
                txn.put('hr_{}_{}'.format(sizes[1], key).encode('utf-8'), hr_img)
                txn.put('sr_{}_{}_{}'.format(sizes[0], sizes[1], key).encode('utf-8'), sr_img)
                txn.put('length'.encode('utf-8'), str(total + 1).encode('utf-8'))
        total += 1
    return total


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--path', '-p', type=str, default='{}/Dataset/celebahq_256'.format(Path.home()))
    parser.add_argument('--out', '-o', type=str, default='./dataset/celebahq')
    parser.add_argument('--size', type=str, default='64,512')
    parser.add_argument('--n_worker', type=int, default=3)
    parser.add_argument('--resample', type=str, default='bicubic')
    parser.add_argument('--lmdb', '-l', action='store_true')
    args = parser.parse_args()
    resample = {'bilinear': Image.BILINEAR, 'bicubic': Image.BICUBIC}[args.resample]
    sizes = [int(s.strip()) for s in args.size.split(',')]
    args.out = '{}_{}_{}'.format(args.out, sizes[0], sizes[1])
    prepare(args.path, args.out, args.n_worker, sizes=sizes, resample=resample, lmdb_save=args.lmdb)
