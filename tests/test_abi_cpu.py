"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/sr3_mi355x.h
and include/sr3_io_mi355x.h declare, the plan's parameter table is the reference's state-dict schema, and the drop-in package
round-trips checkpoints and reproduces the reference's initialisation order."""
import os
import re

import pytest
import torch

from helpers import DESCS, ROOT, load_golden, opt_for


def header_symbols():
    syms = set()
    for h in ('sr3_mi355x.h', 'sr3_io_mi355x.h'):
        src = open(os.path.join(ROOT, 'include', h)).read()
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        syms |= set(re.findall(r'\b(sr3_[a-z0-9_]+)\s*\(', src))
    return sorted(syms)


def test_library_exports_every_declared_symbol():
    from sr3_hip import lib as L
    lib = L.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), s
        assert s in L.SIGNATURES, 'ctypes signature missing for ' + s
    assert set(L.SIGNATURES) == set(syms)
    assert lib.sr3_version() == 1


def test_plan_error_convention_no_gpu():
    from sr3_hip import engine as E, lib as L
    with pytest.raises(L.Sr3Error) as e:
        E.Plan('sr3', 6, 3, 6, 2, [1, 2], [8], 1, 16)          # inner_channel % 4 != 0
    assert 'multiple of 4' in str(e.value)
    with pytest.raises(L.Sr3Error):
        E.Plan('sr3', 6, 3, 8, 4, [1, 2, 2], [8], 1, 18)       # image size not divisible


@pytest.mark.parametrize('name', ['sr3_tiny', 'ddpm_tiny', 'sr3_seam'])
def test_param_table_is_reference_state_dict_schema(name):
    from sr3_hip import engine as E
    d = DESCS[name]
    p = E.Plan(d['variant'], d['in_channel'], d['out_channel'], d['inner_channel'], d['norm_groups'],
               d['channel_mults'], d['attn_res'], d['res_blocks'], d['image_size'])
    g, sd = load_golden(name)
    ref = {k[len('denoise_fn.'):]: tuple(v.shape) for k, v in sd.items()
           if k.startswith('denoise_fn.') and not k.endswith('inv_freq')}
    tab = {e['name']: e['shape'] for e in p.table}
    assert tab == ref
    # arena entries are 16-byte aligned, disjoint and inside the arena
    spans = sorted((e['offset'], e['offset'] + e['numel']) for e in p.table)
    assert all(a % 4 == 0 for a, _ in spans)
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
    assert spans[-1][1] <= p.param_floats


def test_full_size_plan_matches_survey_counts():
    from sr3_hip import engine as E
    p = E.Plan('sr3', 6, 3, 64, 32, [1, 2, 4, 8, 8], [16], 2, 128)
    assert sum(e['numel'] for e in p.table) == 97807491
    assert abs(p.forward_flops(1) / 1e9 - 92.35) < 0.05           # GFLOP / image / forward
    assert p.workspace_bytes(16) < 2 << 30
    q = E.Plan('ddpm', 3, 3, 64, 32, [1, 1, 2, 2, 4, 4], [16], 2, 128)
    assert sum(e["numel"] for e in q.table) == 26449859           # parameters only (inv_freq is a buffer)
    r = E.Plan('sr3', 6, 3, 64, 16, [1, 2, 4, 8, 16], [], 1, 512)
    assert sum(e['numel'] for e in r.table) == 155334339


@pytest.mark.parametrize('name', ['sr3_tiny', 'ddpm_tiny', 'sr3_seam'])
def test_dropin_state_dict_roundtrip(name):
    import model as Model
    m = Model.create_model(opt_for(name, gpu=False))
    g, sd = load_golden(name)
    m.netG.load_state_dict(sd, strict=True)
    out = m.netG.state_dict()
    assert set(out.keys()) == set(sd.keys())
    for k in sd:
        assert torch.equal(out[k].cpu(), sd[k]), k
    with pytest.raises(RuntimeError):
        bad = dict(sd)
        bad.pop('denoise_fn.final_conv.block.3.bias')
        m.netG.load_state_dict(bad, strict=True)


def test_save_and_load_network_files(tmp_path):
    import model as Model
    opt = opt_for('sr3_tiny', phase='train', gpu=False)
    opt['path']['checkpoint'] = str(tmp_path)
    torch.manual_seed(3)
    m = Model.create_model(opt)
    m.save_network(epoch=2, iter_step=7)
    assert os.path.exists(tmp_path / 'I7_E2_gen.pth') and os.path.exists(tmp_path / 'I7_E2_opt.pth')
    opt2 = opt_for('sr3_tiny', phase='train', gpu=False)
    opt2['path']['resume_state'] = str(tmp_path / 'I7_E2')
    torch.manual_seed(99)
    m2 = Model.create_model(opt2)
    a, b = m.netG.state_dict(), m2.netG.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert m2.begin_step == 7 and m2.begin_epoch == 2


def test_schedule_buffers_match_golden():
    import model as Model
    m = Model.create_model(opt_for('sr3_tiny', gpu=False))
    g, sd = load_golden('sr3_tiny')
    for k in ('betas', 'sqrt_recip_alphas_cumprod', 'posterior_log_variance_clipped', 'posterior_mean_coef2'):
        assert torch.equal(getattr(m.netG, k), sd[k]), k
    import numpy as np
    assert np.array_equal(m.netG.sqrt_alphas_cumprod_prev, g['meta/host_sqrt_alphas_cumprod_prev'])


def test_engine_has_no_cpu_fallback():
    import model as Model
    from sr3_hip import lib as L
    m = Model.create_model(opt_for('sr3_tiny', gpu=False))
    with pytest.raises(L.Sr3Error):
        m.netG.denoise_fn(torch.zeros(1, 6, 16, 16), torch.zeros(1, 1))
    m.feed_data({'HR': torch.zeros(1, 3, 16, 16), 'SR': torch.zeros(1, 3, 16, 16)})
    with pytest.raises(L.Sr3Error):
        m.test()


def test_optimizer_state_is_torch_adam_format():
    """`*_opt.pth` interchange: the engine's optimizer state loads into a real
    torch.optim.Adam built over reference-shaped parameters, and back."""
    import model as Model
    opt = opt_for('sr3_tiny', phase='train', gpu=False)
    m = Model.create_model(opt)
    un = m.netG.denoise_fn
    optG = m.optG
    g = torch.Generator().manual_seed(5)
    optG.exp_avg = torch.randn(un.arena.numel(), generator=g)
    optG.exp_avg_sq = torch.rand(un.arena.numel(), generator=g)
    optG.step_count = 7
    sd = optG.state_dict()
    params = [torch.nn.Parameter(p.detach().clone().contiguous()) for p in un.parameters()]
    ref = torch.optim.Adam(params, lr=1e-4)
    ref.load_state_dict(sd)                                  # torch accepts the layout
    rsd = ref.state_dict()
    assert len(rsd['state']) == len(un.plan.table) == 162
    k = [i for i, e in enumerate(un.plan.table) if e['pack'] == 1][3]       # a 3x3 conv weight (OIHW)
    assert tuple(rsd['state'][k]['exp_avg'].shape) == tuple(un.plan.table[k]['shape'])
    # and back: a torch state dict loads into the engine optimizer bit for bit
    optG2 = Model.create_model(opt_for('sr3_tiny', phase='train', gpu=False)).optG
    optG2.load_state_dict(rsd)
    assert optG2.step_count == 7
    for e in un.plan.table:
        assert torch.equal(un.plan.view(optG2.exp_avg, e), un.plan.view(optG.exp_avg, e)), e['name']
        assert torch.equal(un.plan.view(optG2.exp_avg_sq, e), un.plan.view(optG.exp_avg_sq, e))
