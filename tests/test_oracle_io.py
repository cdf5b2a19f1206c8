"""CPU checks of the io / metrics restatement (oracle/io_metrics_oracle.py) against the vectors the reference's own
core/metrics.py and data/util.py produced (tests/golden/io_metrics.npz, oracle/make_golden_io.py), of the restated
cv2 / torchvision pieces against scipy, and of the drop-in dataset's host-side logic (decode only, no GPU)."""
import os

import numpy as np
import pytest
import torch

from helpers import ROOT
from oracle import io_metrics_oracle as O


@pytest.fixture(scope='module')
def g():
    return dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'io_metrics.npz')))


def test_tensor2img_matches_reference(g):
    x = g['t2i/x']
    for key, got in [('single', O.tensor2img(x[:1])), ('chw', O.tensor2img(x[2])), ('grid5', O.tensor2img(x)),
                     ('grid4_01', O.tensor2img(x[:4], min_max=(0, 1))), ('gray_grid', O.tensor2img(g['t2i/gray_x'])),
                     ('gray_2d', O.tensor2img(g['t2i/gray_x'][:1]))]:
        ref = g['t2i/' + key]
        assert got.dtype == np.uint8 and got.shape == ref.shape, key
        assert np.array_equal(got, ref), key
    f = O.tensor2img(x[:1], out_type=np.float32)
    assert f.dtype == np.float32 and np.array_equal(f, g['t2i/float_single'])
    # shapes: 5 images of 12x14 -> nrow 2 -> 3 rows x 2 columns of (14, 16) cells + 2
    assert g['t2i/grid5'].shape == (3 * 14 + 2, 2 * 16 + 2, 3)
    assert g['t2i/gray_grid'].shape[2] == 3 and g['t2i/gray_2d'].ndim == 2


def test_psnr_ssim_match_reference(g):
    a, b = g['m/a'], g['m/b']
    assert O.calculate_psnr(a, b) == float(g['m/psnr'])
    assert O.calculate_psnr(a, a) == float('inf') == float(g['m/psnr_same'])
    assert O.calculate_ssim(a, b) == pytest.approx(float(g['m/ssim3']), rel=1e-13)
    assert O.calculate_ssim(a[:, :, :1], b[:, :, :1]) == pytest.approx(float(g['m/ssim1']), rel=1e-13)
    assert O.calculate_ssim(a[:, :, 0], b[:, :, 0]) == pytest.approx(float(g['m/ssim2d']), rel=1e-13)
    assert O.calculate_psnr(g['m/n'], g['m/s']) == float(g['m/psnr_sn'])
    assert O.calculate_ssim(g['m/n'], g['m/s']) == pytest.approx(float(g['m/ssim_sn']), rel=1e-13)
    assert 0.5 < float(g['m/ssim_sn']) < 1.0 and 20 < float(g['m/psnr_sn']) < 40
    with pytest.raises(ValueError):
        O.calculate_ssim(a, b[:-1])


def test_restated_opencv_pieces_against_scipy():
    """cv2.getGaussianKernel / cv2.filter2D are restated (OpenCV is absent): check them against scipy."""
    from scipy import ndimage, signal
    k = O.gaussian_kernel(11, 1.5)
    w = signal.windows.gaussian(11, 1.5)
    assert k.shape == (11, 1) and np.allclose(k[:, 0], w / w.sum(), rtol=1e-15, atol=0)
    rng = np.random.RandomState(3)
    img = rng.rand(30, 27, 3) * 255
    window = np.outer(k, k.T)
    got = O.filter2d_valid(img, window)
    ref = np.stack([ndimage.correlate(img[:, :, c], window, mode='mirror') for c in range(3)], -1)[5:-5, 5:-5]
    assert got.shape == ref.shape and np.allclose(got, ref, rtol=1e-13, atol=1e-11)


def test_transform_augment_matches_reference(g):
    imgs = [g['tr/in0'], g['tr/in1']]
    r = O.transform_augment(imgs, split='val', min_max=(-1, 1))
    assert np.array_equal(r[0], g['tr/val0']) and np.array_equal(r[1], g['tr/val1'])
    r = O.transform_augment(imgs, split='train', min_max=(-1, 1), flip=True)
    assert np.array_equal(r[0], g['tr/train_flip0']) and np.array_equal(r[1], g['tr/train_flip1'])
    r = O.transform_augment(imgs, split='train', min_max=(0, 1), flip=False)
    assert np.array_equal(r[0], g['tr/train_noflip01_0'])
    assert r[0].dtype == np.float32 and r[0].shape == (3, 8, 9)


def _write_triplets(root, n, l=16, r=32, seed=0):
    from PIL import Image
    rng = np.random.RandomState(seed)
    for sub, s in (('lr_%d' % l, l), ('sr_%d_%d' % (l, r), r), ('hr_%d' % r, r)):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
        for i in range(n):
            Image.fromarray(rng.randint(0, 256, size=(s, s, 3)).astype(np.uint8)).save(os.path.join(root, sub, '%05d.png' % i))


def test_dataset_decodes_to_bytes(tmp_path):
    import data as Data
    from data.LRHR_dataset import LRHRDataset
    from PIL import Image
    root = str(tmp_path / 'ds')
    _write_triplets(root, 5)
    opt = dict(name='t', mode='LRHR', dataroot=root, datatype='img', l_resolution=16, r_resolution=32, data_len=3)
    ds = Data.create_dataset(opt, 'val')
    assert isinstance(ds, LRHRDataset) and len(ds) == 3 and ds.dataset_len == 5
    it = ds[1]
    assert set(it) == {'HR', 'SR', 'LR', 'Index', 'flip'} and it['Index'] == 1 and it['flip'] is False
    assert it['HR'].dtype == torch.uint8 and tuple(it['HR'].shape) == (32, 32, 3) and tuple(it['LR'].shape) == (16, 16, 3)
    ref = np.asarray(Image.open(os.path.join(root, 'hr_32', '00001.png')).convert('RGB'))
    assert np.array_equal(it['HR'].numpy(), ref)
    # split 'train': one flip draw per sample; LR + train raises like the reference's torch.stack
    tr = LRHRDataset(root, 'img', 16, 32, split='train', data_len=-1, need_LR=False)
    assert len(tr) == 5
    torch.manual_seed(0)
    flips = [tr[i]['flip'] for i in range(5)] * 8
    torch.manual_seed(0)
    assert [bool(torch.rand(1).item() < 0.5) for _ in range(5)] == flips[:5]
    with pytest.raises(RuntimeError):
        LRHRDataset(root, 'img', 16, 32, split='train', need_LR=True)[0]
    with pytest.raises(NotImplementedError):
        LRHRDataset(root, 'zip')
    with pytest.raises(AssertionError):
        LRHRDataset(str(tmp_path / 'missing'), 'img')


def test_io_entry_points_fail_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import core.metrics as M
    import data.util as U
    from sr3_hip import lib as L
    with pytest.raises(L.Sr3Error):
        M.tensor2img(torch.zeros(1, 3, 16, 16))
    with pytest.raises(L.Sr3Error):
        M.calculate_psnr(np.zeros((16, 16, 3), np.uint8), np.zeros((16, 16, 3), np.uint8))
    with pytest.raises(L.Sr3Error):
        U.transform_augment([np.zeros((4, 4, 3), np.uint8)])


RESIZE_CASES = [(128, 128, 16, 16), (16, 16, 128, 128), (37, 53, 16, 23), (20, 31, 64, 40), (256, 256, 64, 64),
                (64, 64, 512, 512), (33, 33, 33, 50), (50, 20, 20, 20), (12, 300, 12, 16), (5, 7, 1, 1), (1, 1, 9, 4)]


@pytest.mark.parametrize('cubic', [True, False], ids=['bicubic', 'bilinear'])
def test_pillow_resample_restatement_matches_pillow(cubic):
    """data/prepare_data.py's arithmetic is PIL.Image.resize: pin the restatement against the installed Pillow."""
    from PIL import Image
    rng = np.random.RandomState(1)
    for (h, w, oh, ow) in RESIZE_CASES:
        for ch in (3, 1):
            a = rng.randint(0, 256, size=(h, w, ch)).astype(np.uint8)
            a[: h // 2] = np.where(rng.rand(h // 2, w, ch) < 0.5, 0, 255)           # saturating edges: exercises the clamp
            pil = Image.fromarray(a if ch == 3 else a[:, :, 0])
            ref = np.asarray(pil.resize((ow, oh), Image.BICUBIC if cubic else Image.BILINEAR))
            got = O.pil_resize(a, (oh, ow), cubic)
            assert np.array_equal(got.reshape(ref.shape), ref), (h, w, oh, ow, ch)
