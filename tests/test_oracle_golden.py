"""Pin the CPU oracle (oracle/sr3_oracle.py) against vectors produced by the reference itself
(oracle/make_golden.py, committed under tests/golden/).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import sr3_oracle as O
from helpers import DESCS, SCHEDS, CONDITIONAL, load_golden

NAMES = ['sr3_tiny', 'ddpm_tiny', 'sr3_seam', 'sr3_uncond']
TOL = 2e-6     # same torch CPU ops in a different call order; observed ~1e-7


def close(a, b, tol=TOL):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    d = np.abs(a - b).max()
    assert d <= tol * max(1.0, np.abs(b).max()), d


@pytest.mark.parametrize('name', NAMES)
def test_unet_forward_and_taps(name):
    g, sd = load_golden(name)
    taps = {}
    x = torch.from_numpy(g['unet/x']); t = torch.from_numpy(g['unet/time'])
    with torch.no_grad():
        eps = O.unet_forward(sd, DESCS[name], x, t, taps=taps)
    close(eps.numpy(), g['unet/eps'])
    n = 0
    for k, v in g.items():
        if k.startswith('unet/tap/'):
            close(taps[k[len('unet/tap/'):]].numpy(), v)
            n += 1
    assert n == len(taps) and n > 4


@pytest.mark.parametrize('name', NAMES)
def test_schedule_tables(name):
    g, sd = load_golden(name)
    tab = O.schedule_tables(SCHEDS[name])
    for k in ('betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
              'sqrt_one_minus_alphas_cumprod', 'log_one_minus_alphas_cumprod', 'sqrt_recip_alphas_cumprod',
              'sqrt_recipm1_alphas_cumprod', 'posterior_variance', 'posterior_log_variance_clipped',
              'posterior_mean_coef1', 'posterior_mean_coef2'):
        assert np.array_equal(tab[k], g['sd/' + k]), k
    if DESCS[name]['variant'] == 'sr3':
        assert np.array_equal(tab['sqrt_alphas_cumprod_prev'], g['meta/host_sqrt_alphas_cumprod_prev'])
    assert tab['num_timesteps'] == int(g['meta/T'])


@pytest.mark.parametrize('name', NAMES)
def test_p_sample_steps_and_loop(name):
    g, sd = load_golden(name)
    d = DESCS[name]; tab = O.schedule_tables(SCHEDS[name]); cond = CONDITIONAL[name]
    sr = torch.from_numpy(g['loop/sr']); zs = torch.from_numpy(g['loop/zs'])
    xs = torch.from_numpy(g['step/x'])
    T = tab['num_timesteps']
    with torch.no_grad():
        for t in sorted({T - 1, T // 2, 0}):
            r = O.p_sample(sd, d, tab, xs, t, zs[t], condition_x=sr if cond else None)
            close(r.numpy(), g['step/%d' % t])
        x_T = torch.from_numpy(g['loop/x_T'])
        for cont in (True, False):
            r = O.p_sample_loop(sd, d, tab, sr, x_T, zs, conditional=cond, continous=cont)
            close(r.numpy(), g['loop/ret_continous' if cont else 'loop/ret_last'], 5e-6)


@pytest.mark.parametrize('name', ['sr3_tiny', 'ddpm_tiny'])
def test_p_sample_without_clipping(name):
    """`clip_denoised=False` (sr3 diffusion.py:162-163, ddpm :184-185) against the reference's own outputs on inputs where
    the clamp would bite (tests/golden/noclip.npz, oracle/make_golden_noclip.py)."""
    import os
    from helpers import GOLDEN
    g, sd = load_golden(name)
    n = np.load(os.path.join(GOLDEN, 'noclip.npz'))
    d = DESCS[name]; tab = O.schedule_tables(SCHEDS[name]); cond = CONDITIONAL[name]
    sr = torch.from_numpy(g['loop/sr']); zs = torch.from_numpy(g['loop/zs'])
    x = torch.from_numpy(n[name + '/x'])
    T = tab['num_timesteps']
    with torch.no_grad():
        for t in sorted({T - 1, T // 2, 0}):
            kw = dict(condition_x=sr if cond else None)
            mean = O.p_sample(sd, d, tab, x, t, torch.zeros_like(x), clip_denoised=False, **kw)
            close(mean.numpy(), n['%s/mean/%d' % (name, t)])
            step = O.p_sample(sd, d, tab, x, t, zs[t], clip_denoised=False, **kw)
            close(step.numpy(), n['%s/step/%d' % (name, t)])
            clipped = O.p_sample(sd, d, tab, x, t, torch.zeros_like(x), **kw)
            assert np.abs(clipped.numpy() - n['%s/mean/%d' % (name, t)]).max() > 1e-3       # the switch matters here


@pytest.mark.parametrize('name', ['sr3_tiny', 'ddpm_tiny', 'sr3_uncond'])
def test_p_losses(name):
    g, sd = load_golden(name)
    d = DESCS[name]; tab = O.schedule_tables(SCHEDS[name]); cond = CONDITIONAL[name]
    hr = torch.from_numpy(g['loop/hr']); sr = torch.from_numpy(g['loop/sr']); z = torch.from_numpy(g['train/z'])
    with torch.no_grad():
        if d['variant'] == 'sr3':
            loss = O.p_losses_sr3(sd, d, hr, sr, torch.from_numpy(g['train/gamma']), z, conditional=cond)
        else:
            loss = O.p_losses_ddpm(sd, d, tab, hr, sr, torch.from_numpy(g['train/t']), z, conditional=cond)
    assert abs(float(loss) - float(g['train/loss_sum'])) <= 1e-5 * abs(float(g['train/loss_sum']))
    assert abs(float(loss) / hr.numel() - float(g['train/l_pix'])) <= 1e-5 * abs(float(g['train/l_pix']))


def test_topology_matches_reference_counts():
    """SR3 16->128 has 27 resnet blocks, 6 attention cores, 4 down / 4 up."""
    d = dict(variant='sr3', in_channel=6, out_channel=3, inner_channel=64, norm_groups=32,
             channel_mults=[1, 2, 4, 8, 8], attn_res=[16], res_blocks=2, image_size=128)
    topo = O.unet_topology(d)
    layers = topo['downs'] + topo['mid'] + topo['ups']
    assert sum(l['kind'] == 'res' for l in layers) == 27
    assert sum(l['kind'] == 'res' and l['attn'] for l in layers) == 6
    assert sum(l['kind'] == 'down' for l in layers) == 4 and sum(l['kind'] == 'up' for l in layers) == 4
    assert [l['cin'] for l in topo['ups'] if l['kind'] == 'res'][:4] == [1024, 1024, 1024, 1024]


SCHED_NAMES = ['quad', 'linear', 'warmup10', 'warmup50', 'const', 'jsd', 'cosine']


@pytest.mark.parametrize('sched', SCHED_NAMES)
def test_every_beta_schedule_matches_reference(sched):
    """make_beta_schedule for all seven names (reference diffusion.py:12-49; tests/golden/schedules.npz comes from the
    reference itself, oracle/make_golden_sched.py): the oracle's and the engine's host code, float64, bit for bit."""
    import os
    from helpers import ROOT
    from sr3_hip import diffusion as D
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'schedules.npz'))
    for n in (20, 2000):
        for flavour, (ls, le) in (('sr3', (1e-6, 1e-2)), ('ddpm', (1e-4, 2e-2))):
            ref = g['%s/%s/%d' % (flavour, sched, n)]
            a = np.asarray(O.make_beta_schedule(sched, n, ls, le), dtype=np.float64)
            b = np.asarray(D.make_beta_schedule(sched, n, ls, le), dtype=np.float64)
            assert a.shape == ref.shape and np.array_equal(a, ref), (flavour, n, 'oracle')
            assert b.shape == ref.shape and np.array_equal(b, ref), (flavour, n, 'engine')


def test_cosine_schedule_buffers_match_reference():
    import os
    from helpers import ROOT
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'schedules.npz'))
    opt = dict(schedule='cosine', n_timestep=50, linear_start=1e-6, linear_end=1e-2)
    tab = O.schedule_tables(opt)
    keys = [k[len('buf/cosine50/'):] for k in g.files if k.startswith('buf/cosine50/')]
    assert len(keys) == 13
    for k in keys:
        assert np.array_equal(tab[k], g['buf/cosine50/' + k]), k
    # the engine's module registers the same buffers (CPU construction works without a GPU)
    import model.networks as networks
    from helpers import opt_for
    o = opt_for('sr3_tiny', phase='val', gpu=False)
    netG = networks.define_G(o)
    netG.set_new_noise_schedule(opt, torch.device('cpu'))
    sdn = netG.state_dict()
    for k in keys:
        if k == 'sqrt_alphas_cumprod_prev':
            assert np.array_equal(np.asarray(netG.sqrt_alphas_cumprod_prev), g['buf/cosine50/' + k])
        else:
            assert np.array_equal(sdn[k].numpy(), g['buf/cosine50/' + k]), k


def test_winograd_f2x3_fp32_is_in_the_direct_convolutions_error_class():
    """The engine's 3x3 convolutions run as Winograd F(2x2,3x3) in fp32 (csrc/conv3x3_wino.hip).  CPU restatement of that
    arithmetic (transforms B^T d B, G g G^T, A^T m A with their 0 / +-1 / +-1/2 coefficients, everything fp32) on one layer
    against float64: its error must stay within a small factor of the plain fp32 convolution's -- the numerical basis for
    keeping the stated tolerance unchanged (tools/winograd_numerics.py runs the same emulation over the whole UNet)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 64, 32, 32, generator=g)
    w = torch.randn(96, 64, 3, 3, generator=g) / 24.0
    BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
    AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
    p = F.pad(x, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)                  # B, C, H/2, W/2, 4, 4 patches, stride 2
    V = torch.einsum('ir,bcyxrs,js->bcyxij', BT, p, BT)
    U = torch.einsum('ir,ocrs,js->ocij', G, w, G)
    M = torch.einsum('bcyxij,ocij->boyxij', V, U)
    Y = torch.einsum('pi,boyxij,qj->boyxpq', AT, M, AT).permute(0, 1, 2, 4, 3, 5).reshape(2, 96, 32, 32)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    direct = F.conv2d(x, w, padding=1)
    e_w = (Y.double() - ref).abs().max().item()
    e_d = (direct.double() - ref).abs().max().item()
    assert Y.dtype == torch.float32 and e_w <= 4.0 * e_d + 1e-7 * ref.abs().max().item(), (e_w, e_d)
    assert e_w <= 2e-5 * max(1.0, ref.abs().max().item())
