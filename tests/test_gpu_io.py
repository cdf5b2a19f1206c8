"""GPU parity of the steps either side of the hot path through include/sr3_io_mi355x.h:
tensor2img / PSNR / SSIM (drop-in core/metrics.py) and transform_augment / the device-side batch loader (drop-in
data/) against the reference-generated vectors (tests/golden/io_metrics.npz) and the numpy oracle.
Bars: uint8 images and the PSNR reduction bit-exact; fp32 transform outputs bit-exact; SSIM rel 1e-10 (double, a
different but fixed summation order)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import ROOT                              # noqa: E402
from oracle import io_metrics_oracle as O             # noqa: E402
from test_oracle_io import _write_triplets             # noqa: E402


@pytest.fixture(scope='module')
def g():
    return dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'io_metrics.npz')))


def test_tensor2img_golden(g):
    import core.metrics as M
    x = torch.from_numpy(g['t2i/x'])
    gray = torch.from_numpy(g['t2i/gray_x'])
    cases = [('single', x[:1], {}), ('chw', x[2], {}), ('grid5', x, {}), ('grid4_01', x[:4], dict(min_max=(0, 1))),
             ('gray_grid', gray, {}), ('gray_2d', gray[:1], {})]
    for key, t, kw in cases:
        for dev in ('cpu', 'cuda'):
            got = M.tensor2img(t.to(dev), **kw)
            assert got.dtype == np.uint8 and got.shape == g['t2i/' + key].shape, key
            assert np.array_equal(got, g['t2i/' + key]), key
    f = M.tensor2img(x[:1], out_type=np.float32)
    assert f.dtype == np.float32 and np.array_equal(f, g['t2i/float_single'])
    d = M.tensor2img_device(x[:1].cuda())
    assert d.is_cuda and d.dtype == torch.uint8
    with pytest.raises(TypeError):
        M.tensor2img(torch.zeros(2, 2, 2, 2, 2))


@pytest.mark.parametrize('shape', [(1, 3, 128, 128), (16, 3, 128, 128), (7, 3, 33, 17), (9, 1, 20, 24), (1, 1, 64, 64), (2, 3, 512, 512)])
def test_tensor2img_vs_oracle(shape):
    import core.metrics as M
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=gen) * 0.7
    got = M.tensor2img(x.cuda())
    ref = O.tensor2img(x.numpy())
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_psnr_ssim_golden(g):
    import core.metrics as M
    a, b = g['m/a'], g['m/b']
    assert M.calculate_psnr(a, b) == float(g['m/psnr'])                    # exact: integer SSE, same final expression
    assert M.calculate_psnr(a, a) == float('inf')
    assert M.calculate_ssim(a, b) == pytest.approx(float(g['m/ssim3']), rel=1e-10)
    assert M.calculate_ssim(a[:, :, :1], b[:, :, :1]) == pytest.approx(float(g['m/ssim1']), rel=1e-10)
    assert M.calculate_ssim(a[:, :, 0], b[:, :, 0]) == pytest.approx(float(g['m/ssim2d']), rel=1e-10)
    assert M.calculate_psnr(g['m/n'], g['m/s']) == float(g['m/psnr_sn'])
    assert M.calculate_ssim(g['m/n'], g['m/s']) == pytest.approx(float(g['m/ssim_sn']), rel=1e-10)
    with pytest.raises(ValueError):
        M.calculate_ssim(a, b[:-1])
    with pytest.raises(TypeError):
        M.calculate_psnr(a.astype(np.float64), b.astype(np.float64))
    with pytest.raises(ValueError):
        M.calculate_ssim(a[:8, :8], b[:8, :8])


@pytest.mark.parametrize('hw', [(128, 128), (11, 11), (37, 50), (512, 512)])
def test_psnr_ssim_vs_oracle(hw):
    import core.metrics as M
    rng = np.random.RandomState(hw[0] * 7 + hw[1])
    a = rng.randint(0, 256, size=hw + (3,)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.randint(-30, 31, size=a.shape), 0, 255).astype(np.uint8)
    assert M.calculate_psnr(a, b) == O.calculate_psnr(a, b)
    assert M.calculate_ssim(a, b) == pytest.approx(O.calculate_ssim(a, b), rel=1e-10)
    # device tensors in, nothing but the scalar out
    assert M.calculate_psnr(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()) == O.calculate_psnr(a, b)


def test_fused_validation_metrics_batch():
    """sr.py:119-145 per-image chain for a whole (16, 3, 128, 128) batch: one call, 2 scalars per image back."""
    import core.metrics as M
    gen = torch.Generator().manual_seed(9)
    hr = (torch.rand(16, 3, 128, 128, generator=gen) * 2 - 1)
    sr = (hr + 0.08 * torch.randn(16, 3, 128, 128, generator=gen))        # leaves [-1, 1]: exercises the clamp
    sr[3] = hr[3]                                                            # identical pair -> inf
    psnr, ssim = M.psnr_ssim_batch(sr.cuda(), hr.cuda())
    assert len(psnr) == 16 and len(ssim) == 16
    for i in range(16):
        a = O.tensor2img(sr[i].numpy())
        b = O.tensor2img(hr[i].numpy())
        assert psnr[i] == O.calculate_psnr(a, b), i
        assert ssim[i] == pytest.approx(O.calculate_ssim(a, b), rel=1e-10), i
    assert psnr[3] == float('inf') and ssim[3] == pytest.approx(1.0, rel=1e-12)
    p1, s1 = M.psnr_ssim_batch(sr[5], hr[5])                                 # 3-D (C, H, W) inputs, host tensors
    assert p1 == [psnr[5]] and s1 == [ssim[5]]


def test_transform_augment_golden_and_oracle(g):
    import data.util as U
    imgs = [g['tr/in0'], g['tr/in1']]
    r = U.transform_augment(imgs, split='val', min_max=(-1, 1))
    assert r[0].is_cuda and r[0].dtype == torch.float32 and tuple(r[0].shape) == (3, 8, 9)
    assert np.array_equal(r[0].cpu().numpy(), g['tr/val0']) and np.array_equal(r[1].cpu().numpy(), g['tr/val1'])
    u8 = torch.from_numpy(np.stack(imgs, 0))
    out = U.u8_batch_to_f32(u8, [True, True], (-1, 1))
    assert np.array_equal(out[0].cpu().numpy(), g['tr/train_flip0']) and np.array_equal(out[1].cpu().numpy(), g['tr/train_flip1'])
    out = U.u8_batch_to_f32(u8, [False, True], (0, 1))
    assert np.array_equal(out[0].cpu().numpy(), g['tr/train_noflip01_0'])
    assert np.array_equal(out[1].cpu().numpy(), O.transform_augment([imgs[1]], 'train', (0, 1), flip=True)[0])
    # split 'train': the pair is flipped together or not at all
    torch.manual_seed(4)
    seen = set()
    for _ in range(12):
        r = U.transform_augment(imgs, split='train', min_max=(-1, 1))
        f0 = np.array_equal(r[0].cpu().numpy(), g['tr/train_flip0'])
        f1 = np.array_equal(r[1].cpu().numpy(), g['tr/train_flip1'])
        assert f0 == f1
        assert f0 or np.array_equal(r[0].cpu().numpy(), g['tr/val0'])
        seen.add(f0)
    assert seen == {True, False}
    # every byte value, full size
    big = torch.arange(256, dtype=torch.uint8).repeat(3 * 128 * 128 * 4 // 256).view(4, 128, 128, 3)
    got = U.u8_batch_to_f32(big, [True, False, True, False], (-1, 1)).cpu().numpy()
    for i in range(4):
        assert np.array_equal(got[i], O.transform_augment([big[i].numpy()], 'train', (-1, 1), flip=(i % 2 == 0))[0])


def test_device_batches_feed_the_model(tmp_path):
    """create_dataset -> create_dataloader -> DDPM.feed_data: the batch dict is the reference's, already on the GPU."""
    import data as Data
    from PIL import Image
    root = str(tmp_path / 'ds')
    _write_triplets(root, 6, l=16, r=32)
    opt = dict(name='t', mode='HR', dataroot=root, datatype='img', l_resolution=16, r_resolution=32, data_len=-1,
               batch_size=4, use_shuffle=False, num_workers=0)
    ds = Data.create_dataset(opt, 'train')
    torch.manual_seed(21)
    batches = list(Data.create_dataloader(ds, opt, 'train'))
    assert len(batches) == 2 and set(batches[0]) == {'HR', 'SR', 'Index'}
    assert tuple(batches[0]['HR'].shape) == (4, 3, 32, 32) and batches[0]['HR'].is_cuda and batches[1]['SR'].shape[0] == 2
    assert batches[0]['Index'].tolist() == [0, 1, 2, 3]
    k = 0
    flips = []
    for bt in batches:
        for j in range(bt['HR'].shape[0]):
            flag = {}
            for key, sub in (('HR', 'hr_32'), ('SR', 'sr_16_32')):
                img = np.asarray(Image.open(os.path.join(root, sub, '%05d.png' % k)).convert('RGB'))
                got = bt[key][j].cpu().numpy()
                hits = [f for f in (False, True) if np.array_equal(got, O.transform_augment([img], 'train', (-1, 1), flip=f)[0])]
                assert len(hits) == 1, (k, key)
                flag[key] = hits[0]
            assert flag['HR'] == flag['SR'], k            # one draw per sample, shared by the pair
            flips.append(flag['HR'])
            k += 1
    assert len(flips) == 6
    # phase 'val': batch 1, LR included, no flips
    vopt = dict(opt, mode='LRHR')
    vds = Data.create_dataset(vopt, 'val')
    vb = next(iter(Data.create_dataloader(vds, vopt, 'val')))
    # (the two underscore entries are the validation wave the loader attaches, sr3_hip.dist.ValWave: DDPM.test runs the chains of
    #  consecutive items as one batch; SR3_VAL_CHAIN_BATCH=1 leaves the reference's dict as it is)
    assert set(vb) == {'HR', 'SR', 'LR', 'Index', '_dp_wave', '_dp_pos'} and tuple(vb['LR'].shape) == (1, 3, 16, 16)
    img = np.asarray(Image.open(os.path.join(root, 'lr_16', '00000.png')).convert('RGB'))
    assert np.array_equal(vb['LR'][0].cpu().numpy(), O.transform_augment([img], 'val', (-1, 1))[0])


def _tv_resize_center_crop(pil, size, resample):
    """torchvision.transforms.functional.resize(img, int) + center_crop(img, int), restated with PIL calls only."""
    w, h = pil.size
    if not ((w <= h and w == size) or (h <= w and h == size)):
        ow, oh = (size, int(size * h / w)) if w < h else (int(size * w / h), size)
        pil = pil.resize((ow, oh), resample)
    w, h = pil.size
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return pil.crop((left, top, left + size, top + size))


@pytest.mark.parametrize('cubic', [True, False], ids=['bicubic', 'bilinear'])
def test_resize_batch_is_pillow_bit_exact(cubic):
    import data.prepare_data as P
    from PIL import Image
    from test_oracle_io import RESIZE_CASES
    rs = Image.BICUBIC if cubic else Image.BILINEAR
    rng = np.random.RandomState(2)
    for (h, w, oh, ow) in RESIZE_CASES + [(512, 512, 64, 64), (1024, 1024, 128, 128), (100, 160, 512, 300)]:
        n = 3
        a = rng.randint(0, 256, size=(n, h, w, 3)).astype(np.uint8)
        a[0, : h // 2] = np.where(rng.rand(h // 2, w, 3) < 0.5, 0, 255)
        got = P.resize_batch(torch.from_numpy(a), (oh, ow), rs).cpu().numpy()
        for i in range(n):
            ref = np.asarray(Image.fromarray(a[i]).resize((ow, oh), rs))
            assert np.array_equal(got[i], ref), (h, w, oh, ow, i)
    g1 = rng.randint(0, 256, size=(2, 40, 30, 1)).astype(np.uint8)
    got = P.resize_batch(torch.from_numpy(g1).cuda(), (16, 16), rs).cpu().numpy()
    assert np.array_equal(got[1, :, :, 0], np.asarray(Image.fromarray(g1[1, :, :, 0]).resize((16, 16), rs)))
    with pytest.raises(NotImplementedError):
        P.resize_batch(torch.from_numpy(g1), (8, 8), Image.NEAREST)


def test_prepare_data_triplets_match_the_reference_pipeline(tmp_path):
    """resize_multiple / prepare (data/prepare_data.py:17-40, 88-157): LR, HR, SR files byte-identical to what
    torchvision resize + center_crop on Pillow produce."""
    import data.prepare_data as P
    from PIL import Image
    rng = np.random.RandomState(5)
    src = tmp_path / 'src'
    os.makedirs(src)
    shapes = [(256, 256), (300, 200), (180, 260), (128, 128)]
    for i, (h, w) in enumerate(shapes):
        yy, xx = np.mgrid[0:h, 0:w]
        base = 127 + 90 * np.sin(yy / 9.0 + i)[:, :, None] * np.cos(xx / 13.0)[:, :, None] * np.ones((1, 1, 3))
        Image.fromarray(np.clip(base + rng.randn(h, w, 3) * 20, 0, 255).astype(np.uint8)).save(str(src / ('%d.png' % i)))
    out = str(tmp_path / 'out_16_128')
    assert P.prepare(str(src), out, n_worker=3, sizes=(16, 128), resample=Image.BICUBIC) == 4
    for i in range(4):
        img = Image.open(str(src / ('%d.png' % i))).convert('RGB')
        lr = _tv_resize_center_crop(img, 16, Image.BICUBIC)
        hr = _tv_resize_center_crop(img, 128, Image.BICUBIC)
        sr = _tv_resize_center_crop(lr, 128, Image.BICUBIC)
        for sub, ref in (('lr_16', lr), ('hr_128', hr), ('sr_16_128', sr)):
            got = np.asarray(Image.open(os.path.join(out, sub, '%05d.png' % i)))
            assert got.shape == np.asarray(ref).shape and np.array_equal(got, np.asarray(ref)), (i, sub)
    # the prepared directory feeds the drop-in dataset
    import data as Data
    ds = Data.create_dataset(dict(name='p', mode='LRHR', dataroot=out, datatype='img', l_resolution=16, r_resolution=128, data_len=-1), 'val')
    assert len(ds) == 4 and tuple(ds[0]['SR'].shape) == (128, 128, 3)
    lr_b, hr_b, sr_b = P.resize_multiple(Image.open(str(src / '0.png')).convert('RGB'), (16, 128), Image.BICUBIC, lmdb_save=True)
    assert np.array_equal(np.asarray(Image.open(__import__('io').BytesIO(sr_b))), np.asarray(Image.open(os.path.join(out, 'sr_16_128', '00000.png'))))


def test_infer_py_call_sequence(tmp_path, monkeypatch):
    """(SR3_VAL_CHAIN_BATCH=1: one chain per image, as the reference runs them -- this test replays each chain's own RNG draws.)
    The call sequence of the reference's infer.py (lines 44-90) through the drop-in packages only: create_dataset /
    create_dataloader('val') on PNG triplets -> Model.create_model -> set_new_noise_schedule(val) -> per image
    feed_data / test(continous=True) / get_current_visuals(need_LR=False) -> Metrics.tensor2img / save_img of HR, INF, the
    SR process grid and the final SR -- and the final image equals the oracle's reverse loop fed the same draws."""
    import core.metrics as M
    import data as Data
    import model as Model
    from PIL import Image
    from helpers import DESCS, SCHEDS, load_golden, opt_for
    from oracle import sr3_oracle as SO
    # (the reference's validation loader forks one worker; forking THIS process -- hundreds of GPU tests' worth of HIP mappings --
    #  takes tens of seconds per pass on the GPU box: load in-process here, the batches are the same)
    import torch.utils.data as _tud
    _DL = _tud.DataLoader
    monkeypatch.setattr(_tud, 'DataLoader', lambda *a, **k: _DL(*a, **dict(k, num_workers=0)))
    monkeypatch.setenv('SR3_VAL_CHAIN_BATCH', '1')
    root = str(tmp_path / 'ds')
    _write_triplets(root, 2, l=4, r=16)
    dopt = dict(name='t', mode='LRHR', dataroot=root, datatype='img', l_resolution=4, r_resolution=16, data_len=-1)
    val_set = Data.create_dataset(dopt, 'val')
    val_loader = Data.create_dataloader(val_set, dopt, 'val')
    opt = opt_for('sr3_tiny', phase='val', gpu=True)
    diffusion = Model.create_model(opt)
    _, sd = load_golden('sr3_tiny')
    diffusion.netG.load_state_dict(sd, strict=True)
    diffusion.netG.show_progress = False
    diffusion.set_new_noise_schedule(opt['model']['beta_schedule']['val'], schedule_phase='val')
    result_path = str(tmp_path / 'results')
    os.makedirs(result_path, exist_ok=True)
    T = SCHEDS['sr3_tiny']['n_timestep']
    tab = SO.schedule_tables(SCHEDS['sr3_tiny'])
    idx = 0
    for _, val_data in enumerate(val_loader):
        idx += 1
        diffusion.feed_data(val_data)
        torch.manual_seed(100 + idx)
        diffusion.test(continous=True)
        visuals = diffusion.get_current_visuals(need_LR=False)
        hr_img = M.tensor2img(visuals['HR'])
        fake_img = M.tensor2img(visuals['INF'])
        sr_img = M.tensor2img(visuals['SR'])
        last = M.tensor2img(visuals['SR'][-1])
        M.save_img(sr_img, '{}/{}_{}_sr_process.png'.format(result_path, 0, idx))
        M.save_img(last, '{}/{}_{}_sr.png'.format(result_path, 0, idx))
        M.save_img(hr_img, '{}/{}_{}_hr.png'.format(result_path, 0, idx))
        M.save_img(fake_img, '{}/{}_{}_inf.png'.format(result_path, 0, idx))
        n_snap = sum(1 for i in range(T) if i % (1 | (T // 10)) == 0)
        assert tuple(visuals['SR'].shape) == (n_snap + 1, 3, 16, 16) and visuals['SR'].device.type == 'cpu'
        assert hr_img.shape == (16, 16, 3) and hr_img.dtype == np.uint8 and last.shape == (16, 16, 3)
        # the saved HR / INF are the dataset's PNGs again (uint8 -> [-1, 1] -> uint8 is the identity)
        k = idx - 1
        assert np.array_equal(np.asarray(Image.open('{}/{}_{}_hr.png'.format(result_path, 0, idx)).convert('RGB')),
                              np.asarray(Image.open(os.path.join(root, 'hr_16', '%05d.png' % k)).convert('RGB')))
        assert np.array_equal(fake_img, np.asarray(Image.open(os.path.join(root, 'sr_4_16', '%05d.png' % k)).convert('RGB')))
        # same chain on the CPU oracle: replay the device RNG draws (x_T, then one z per step) of this seed
        torch.manual_seed(100 + idx)
        d = torch.device('cuda:0')
        x_T = torch.randn((1, 3, 16, 16), device=d)
        zs = [torch.empty((1, 3, 16, 16), device=d).normal_() for _ in range(T)]      # the loop draws at i = T-1 .. 0
        zseq = torch.stack(list(reversed(zs))).cpu()                                    # zs[i] = noise consumed at step i
        with torch.no_grad():
            ref = SO.p_sample_loop(sd, DESCS['sr3_tiny'], tab, val_data['SR'].cpu(), x_T.cpu(), zseq, conditional=True,
                                   continous=True)
        assert tuple(ref.shape) == tuple(visuals['SR'].shape)
        err = (visuals['SR'] - ref).abs().max().item()
        assert err <= 1e-4, err
    assert idx == 2 and len(os.listdir(result_path)) == 8
