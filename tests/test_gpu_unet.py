"""UNet / reverse-step / reverse-loop parity of the engine (through the drop-in `model` package and
the C ABI) against the committed reference vectors (tests/golden) and the CPU oracle.

Stated fp32 tolerances: one forward 2e-5 * max(1, |ref|_inf); full loop 1e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import DESCS, SCHEDS, CONDITIONAL, load_golden, opt_for      # noqa: E402
import gpu_util as G                                                     # noqa: E402

NAMES = ['sr3_tiny', 'ddpm_tiny', 'sr3_seam', 'sr3_uncond']


def build(name, **plan_opts):
    import model as Model
    opt = opt_for(name, phase='val', gpu=True)
    m = Model.create_model(opt)
    g, sd = load_golden(name)
    m.netG.load_state_dict(sd, strict=True)
    for k, v in plan_opts.items():
        m.netG.denoise_fn.plan.set_option(k, v)
    m.netG.show_progress = False
    return m, g, sd


@pytest.mark.parametrize('name', NAMES)
@pytest.mark.parametrize('fuse', [0, 1])
def test_unet_forward_and_layer_taps(name, fuse):
    m, g, sd = build(name, keep_all=1, fuse_stats=fuse)
    un = m.netG.denoise_fn
    d = G.dev()
    x = torch.from_numpy(g['unet/x']).to(d)
    t = torch.from_numpy(g['unet/time']).to(d)
    eps = un(x, t)
    torch.cuda.synchronize()
    ws = un._ws.buf
    off0 = (-ws.data_ptr()) % 256
    worst = 0.0
    for (tname, off, C, H, W) in un.plan.taps():
        B = x.shape[0]
        raw = ws[off0 + off: off0 + off + B * H * W * C * 4].view(torch.float32).view(B, H, W, C)
        got = raw.permute(0, 3, 1, 2).cpu()
        ref = torch.from_numpy(g['unet/tap/' + tname])
        worst = max(worst, G.assert_close(got, ref, what='%s tap %s' % (name, tname)))
    G.assert_close(eps.cpu(), torch.from_numpy(g['unet/eps']), what=name + ' eps')


@pytest.mark.parametrize('name', ['sr3_tiny', 'ddpm_tiny'])
def test_fork_side_option_is_bit_identical_eager_and_captured(name):
    """Plan option fork_side (res_conv and the embedding MLP on the plan's side stream, forked / joined by events): the same kernels on the
    same operands, so the forward is bit-identical to the serial plan's -- launched eagerly, and as the captured graph of the reverse loop
    (where the pairs become parallel branches), under the same seed."""
    m, g, sd = build(name)
    un = m.netG.denoise_fn
    d = G.dev()
    x = torch.from_numpy(g['unet/x']).to(d)
    t = torch.from_numpy(g['unet/time']).to(d)

    def loop():
        torch.manual_seed(17)
        if CONDITIONAL[name]:
            return m.netG.super_resolution(torch.from_numpy(g['unet/x'][:, :3]).to(d), True).clone()
        return m.netG.sample(2, True).clone()
    e0 = un(x, t).clone()
    l0 = loop()
    un.plan.set_option('fork_side', 1)
    ops = un.plan.op_list(x.shape[0])
    assert sum(1 for o in ops if o.get('side_id', -1) >= 0) >= 2, 'nothing was forked'
    for _ in range(3):                       # (repeated: a missing join shows as a race, not every time)
        assert torch.equal(un(x, t), e0)
    assert m.netG.use_graph
    l1 = loop()
    assert torch.equal(l1, l0)
    assert torch.isfinite(l1).all()


@pytest.mark.experiments
@pytest.mark.parametrize('name', NAMES)
def test_unet_forward_split_bf16_option(name):
    """Opt-in `split_bf16` plan option (3 x bf16 operand split on the bf16 MFMA): same stated tolerance."""
    m, g, sd = build(name, split_bf16=1)
    un = m.netG.denoise_fn
    d = G.dev()
    eps = un(torch.from_numpy(g['unet/x']).to(d), torch.from_numpy(g['unet/time']).to(d))
    G.assert_close(eps.cpu(), torch.from_numpy(g['unet/eps']), what=name + ' eps (split_bf16)')
    # the reverse loop, eager and graph replay
    cond = CONDITIONAL[name]
    sr = torch.from_numpy(g['loop/sr']).to(d)
    x_T = torch.from_numpy(g['loop/x_T']).to(d)
    zs = torch.from_numpy(g['loop/zs']).to(d)
    r = m.netG.p_sample_loop(sr if cond else tuple(x_T.shape), continous=True, x_T=x_T, noise_seq=zs)
    G.assert_close(r.cpu(), torch.from_numpy(g['loop/ret_continous']), tol=1e-4, what=name + ' loop (split_bf16)')
    outs = []
    for use_graph in (False, True):
        m.netG.use_graph = use_graph
        torch.manual_seed(7)
        outs.append(m.netG.p_sample_loop(sr if cond else tuple(x_T.shape), continous=False).clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('name', NAMES)
def test_unet_forward_exact_fp32_option(name):
    """Plan options `wino_split` / `gemm_split` (default 1: the Winograd and im2col kernels' 3 x bf16 split instantiations)
    switched off -- every conv on the fp32 MFMA -- on the reference-generated vectors: same stated tolerance; toggling them
    back restores the default plan's result bit for bit."""
    m, g, sd = build(name)
    un = m.netG.denoise_fn
    d = G.dev()
    x, t = torch.from_numpy(g['unet/x']).to(d), torch.from_numpy(g['unet/time']).to(d)
    assert un.plan.options.get('wino_split', 1) == 1 and un.plan.options.get('gemm_split', 1) == 1
    split_tiles = (12, 13, 14, 15, 16, 17, 18, 19, 20, 21)
    has_split = any(o['tile_cfg'] in split_tiles for o in un.plan.op_list(x.shape[0]))
    assert has_split, 'the default plan of %s has no conv on a split instantiation' % name
    e0 = un(x, t).clone()
    G.assert_close(e0.cpu(), torch.from_numpy(g['unet/eps']), what=name + ' eps (default plan)')
    un.plan.set_option('wino_split', 0)
    un.plan.set_option('gemm_split', 0)
    assert not any(o['tile_cfg'] in split_tiles for o in un.plan.op_list(x.shape[0]))
    e1 = un(x, t).clone()
    G.assert_close(e1.cpu(), torch.from_numpy(g['unet/eps']), what=name + ' eps (wino_split = gemm_split = 0)')
    print('%s: eps max abs diff default plan vs all-fp32-MFMA plan %.2e' % (name, float((e0 - e1).abs().max())))
    un.plan.set_option('wino_split', 1)
    un.plan.set_option('gemm_split', 1)
    assert torch.equal(un(x, t), e0)


@pytest.mark.parametrize('name', NAMES)
def test_unet_forward_buffer_reuse_matches(name):
    """The liveness-planned workspace (buffers recycled) gives the same eps as keep_all."""
    m, g, sd = build(name)
    un = m.netG.denoise_fn
    d = G.dev()
    x = torch.from_numpy(g['unet/x']).to(d)
    t = torch.from_numpy(g['unet/time']).to(d)
    e1 = un(x, t).clone()
    if DESCS[name]['in_channel'] == 6:      # conditioning passed separately == pre-concatenated
        e2 = un(x[:, 3:].contiguous(), t, cond=x[:, :3].contiguous())
        assert torch.equal(e1, e2)
    G.assert_close(e1.cpu(), torch.from_numpy(g['unet/eps']), what=name)


@pytest.mark.parametrize('name', NAMES)
def test_p_sample_steps(name):
    m, g, sd = build(name)
    d = G.dev()
    cond = CONDITIONAL[name]
    sr = torch.from_numpy(g['loop/sr']).to(d)
    zs = torch.from_numpy(g['loop/zs']).to(d)
    xs = torch.from_numpy(g['step/x']).to(d)
    T = int(g['meta/T'])
    for t in sorted({T - 1, T // 2, 0}):
        if DESCS[name]['variant'] == 'sr3':
            r = m.netG.p_sample(xs, t, condition_x=sr if cond else None, noise=zs[t])
        else:
            r = m.netG.p_sample(xs, torch.full((xs.shape[0],), t, dtype=torch.long, device=d),
                                condition_x=sr if cond else None, noise=zs[t])
        G.assert_close(r.cpu(), torch.from_numpy(g['step/%d' % t]), what='%s step %d' % (name, t))
    assert torch.equal(xs.cpu(), torch.from_numpy(g['step/x']))     # input untouched


@pytest.mark.parametrize('name', ['sr3_tiny', 'ddpm_tiny'])
def test_p_sample_without_clipping(name):
    """The reference's `clip_denoised=False` branch (sr3 diffusion.py:162-163, ddpm :184-185) through the drop-in's
    p_mean_variance / p_sample, against the reference's own outputs (tests/golden/noclip.npz) on inputs where the clamp bites."""
    import os
    import numpy as np
    from helpers import GOLDEN
    m, g, sd = build(name)
    d = G.dev()
    n = np.load(os.path.join(GOLDEN, 'noclip.npz'))
    cond = CONDITIONAL[name]
    sr = torch.from_numpy(g['loop/sr']).to(d)
    zs = torch.from_numpy(g['loop/zs']).to(d)
    x = torch.from_numpy(n[name + '/x']).to(d)
    T = int(g['meta/T'])
    for t in sorted({T - 1, T // 2, 0}):
        tt = t if DESCS[name]['variant'] == 'sr3' else torch.full((x.shape[0],), t, dtype=torch.long, device=d)
        kw = dict(condition_x=sr) if cond else {}
        ret = m.netG.p_mean_variance(x=x, t=tt, clip_denoised=False, **kw)
        assert len(ret) == (2 if DESCS[name]['variant'] == 'sr3' else 3)          # the DDPM class also returns the variance
        scale = max(1.0, float(np.abs(n['%s/mean/%d' % (name, t)]).max()))
        G.assert_close(ret[0].cpu(), torch.from_numpy(n['%s/mean/%d' % (name, t)]), tol=2e-5 * scale, what='%s mean %d' % (name, t))
        r = m.netG.p_sample(x, tt, clip_denoised=False, noise=zs[t], **kw)
        G.assert_close(r.cpu(), torch.from_numpy(n['%s/step/%d' % (name, t)]), tol=2e-5 * scale, what='%s step %d' % (name, t))
        clipped = m.netG.p_mean_variance(x=x, t=tt, clip_denoised=True, **kw)[0]
        assert (clipped.cpu() - torch.from_numpy(n['%s/mean/%d' % (name, t)])).abs().max() > 1e-3


@pytest.mark.parametrize('clip', [True, False])
@pytest.mark.parametrize('name', NAMES)
def test_reverse_step_one_call_equals_three(name, clip):
    """sr3_reverse_step (the whole loop iteration as one capturable call: the p_sample update and the counter decrement inside the
    output conv's kernel) against sr3_unet_forward + sr3_p_sample_step_ex + sr3_step_decrement on the same inputs: eps and the new
    image BIT-equal, the counter decremented; and against the reference's own step outputs where the golden file holds them."""
    m, g, sd = build(name)
    d = G.dev()
    netG = m.netG
    un = netG.denoise_fn
    cond = torch.from_numpy(g['loop/sr']).to(d) if CONDITIONAL[name] else None
    zs = torch.from_numpy(g['loop/zs']).to(d)
    xs = torch.from_numpy(g['step/x']).to(d)
    T = int(g['meta/T'])
    tables = (netG.sqrt_recip_alphas_cumprod, netG.sqrt_recipm1_alphas_cumprod, netG.posterior_mean_coef1,
              netG.posterior_mean_coef2, netG._sigma)
    for t in sorted({T - 1, T // 2, 1, 0}):
        # three calls
        step1 = torch.full((1,), t, dtype=torch.int32, device=d)
        eps3 = un(xs, None, cond=cond, level_table=netG._level_table, step_dev=step1)
        x3 = xs.clone()
        netG._step_update(x3, eps3, zs[t], step_dev=step1, clip_denoised=clip)
        # one call
        step2 = torch.tensor([-77, t], dtype=torch.int32, device=d)
        x1 = xs.clone()
        eps1 = torch.empty_like(eps3)
        un.reverse_step(x1, zs[t], tables, step2, cond=cond, level_table=netG._level_table, clip_denoised=clip, eps_out=eps1)
        assert torch.equal(eps1, eps3), 'eps differs at t = %d' % t
        assert torch.equal(x1, x3), 'image differs at t = %d' % t
        assert step2.tolist() == [t, t - 1]
        if clip and ('step/%d' % t) in g:
            G.assert_close(x1.cpu(), torch.from_numpy(g['step/%d' % t]), what='%s reverse step %d' % (name, t))
        # eps_out is optional
        x0 = xs.clone()
        step2 = torch.tensor([0, t], dtype=torch.int32, device=d)
        un.reverse_step(x0, zs[t], tables, step2, cond=cond, level_table=netG._level_table, clip_denoised=clip)
        assert torch.equal(x0, x3)


@pytest.mark.parametrize('name', NAMES)
def test_reverse_loop_injected_noise(name):
    m, g, sd = build(name)
    d = G.dev()
    cond = CONDITIONAL[name]
    sr = torch.from_numpy(g['loop/sr']).to(d)
    x_T = torch.from_numpy(g['loop/x_T']).to(d)
    zs = torch.from_numpy(g['loop/zs']).to(d)
    for cont in (True, False):
        arg = sr if cond else tuple(x_T.shape)
        r = m.netG.p_sample_loop(arg, continous=cont, x_T=x_T, noise_seq=zs)
        ref = torch.from_numpy(g['loop/ret_continous' if cont else 'loop/ret_last'])
        assert tuple(r.shape) == tuple(ref.shape)
        G.assert_close(r.cpu(), ref, tol=1e-4, what='%s loop cont=%s' % (name, cont))


@pytest.mark.parametrize('name', ['sr3_tiny', 'ddpm_tiny', 'sr3_uncond'])
def test_graph_replay_equals_eager(name):
    """hipGraph replay of the step (device-side counter, in-graph RNG) == the eager loop, same seed."""
    m, g, sd = build(name)
    d = G.dev()
    cond = CONDITIONAL[name]
    sr = torch.from_numpy(g['loop/sr']).to(d)
    arg = sr if cond else (2, 3, 16, 16)
    outs = []
    for use_graph in (False, True, True):
        m.netG.use_graph = use_graph
        torch.manual_seed(123)
        outs.append(m.netG.p_sample_loop(arg, continous=True).clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    assert torch.isfinite(outs[0]).all()


def test_api_surface_test_and_visuals():
    """feed_data -> test(continous) -> get_current_visuals, with the reference's shape quirks."""
    m, g, sd = build('sr3_tiny')
    B = 2
    data = {'HR': torch.from_numpy(g['loop/hr']), 'SR': torch.from_numpy(g['loop/sr']), 'Index': torch.arange(B)}
    m.feed_data(data)
    m.test(continous=True)
    T = int(g['meta/T'])
    n_snap = sum(1 for i in range(T) if i % (1 | (T // 10)) == 0)
    assert tuple(m.SR.shape) == (B * (n_snap + 1), 3, 16, 16)
    vis = m.get_current_visuals(need_LR=False)
    assert set(vis.keys()) == {'SR', 'INF', 'HR', 'LR'} and vis['SR'].device.type == 'cpu'
    m.test(continous=False)
    assert tuple(m.SR.shape) == (3, 16, 16)       # ret_img[-1]: last image of the batch only


def test_unconditional_sr3_sample_and_visuals():
    """config/sample_sr3_128.json's case (which_model_G sr3, conditional false) through DDPM.sample, as sample.py:104,140
    calls it: the snapshots start from x_T, `continous=False` returns ret_img[-1] -- the LAST image of the batch only,
    (3, H, W) (sr3 diffusion.py:180-187) -- and get_current_visuals(sample=True) returns {'SAM'}."""
    m, g, sd = build('sr3_uncond')
    d = G.dev()
    T = int(g['meta/T'])
    n_snap = sum(1 for i in range(T) if i % (1 | (T // 10)) == 0)
    x_T = torch.from_numpy(g['loop/x_T']).to(d)
    zs = torch.from_numpy(g['loop/zs']).to(d)
    full = m.netG.p_sample_loop(tuple(x_T.shape), continous=True, x_T=x_T, noise_seq=zs)
    ref = torch.from_numpy(g['loop/ret_continous'])
    assert tuple(full.shape) == tuple(ref.shape) == (2 * (n_snap + 1), 3, 16, 16)
    assert torch.equal(full[:2].cpu(), torch.from_numpy(g['loop/x_T']))          # ret_img starts as x_T itself
    G.assert_close(full.cpu(), ref, tol=1e-4, what='unconditional sr3 loop')
    last = m.netG.p_sample_loop(tuple(x_T.shape), continous=False, x_T=x_T, noise_seq=zs)
    assert tuple(last.shape) == (3, 16, 16)
    G.assert_close(last.cpu(), torch.from_numpy(g['loop/ret_last']), tol=1e-4, what='unconditional sr3 ret_img[-1]')
    assert torch.equal(last, full[-1])
    # the wrapper: DDPM.sample(batch_size, continous) -> self.SR, visuals under 'SAM' (model/model.py:68-78, 98-110)
    torch.manual_seed(5)
    m.sample(batch_size=2, continous=True)
    assert tuple(m.SR.shape) == (2 * (n_snap + 1), 3, 16, 16)
    vis = m.get_current_visuals(sample=True)
    assert set(vis.keys()) == {'SAM'} and vis['SAM'].device.type == 'cpu' and tuple(vis['SAM'].shape) == tuple(m.SR.shape)
    torch.manual_seed(5)
    m.sample(batch_size=2, continous=False)
    assert tuple(m.SR.shape) == (3, 16, 16) and torch.equal(m.SR.cpu(), vis['SAM'][-1])


def test_engine_refuses_cpu():
    from sr3_hip import lib as L
    m, g, sd = build('sr3_tiny')
    with pytest.raises(L.Sr3Error):
        m.netG.denoise_fn(torch.zeros(1, 6, 16, 16), torch.zeros(1, 1))


def test_stale_derived_filters_fail_loudly_and_option_toggle_rebuilds():
    """The Winograd filters live in a derived buffer.  (1) The C ABI refuses a forward on filters it was told are
    stale (sr3_plan_invalidate_derived: what the fused Adam step's caller does) or that were prepared from another arena,
    instead of silently computing with the previous weights.  (2) Toggling a plan option that changes the buffer's content
    (winograd off and on again) AFTER a forward keeps the output on the golden eps (the cache key of
    EngineUNet.ensure_derived includes the plan generation)."""
    import ctypes as C
    from sr3_hip import lib as L, engine as E
    m, g, sd = build('sr3_seam')
    un = m.netG.denoise_fn
    d = G.dev()
    x = torch.from_numpy(g['unet/x']).to(d)
    t = torch.from_numpy(g['unet/time']).to(d)
    ref = torch.from_numpy(g['unet/eps'])
    G.assert_close(un(x, t).cpu(), ref, what='before')
    assert any(o['tile_cfg'] in (11, 12, 13) for o in un.plan.op_list(x.shape[0])), 'the plan has no Winograd op: nothing derived to test'
    # (1) raw C-ABI call after an invalidation: loud failure; after prepare: fine again
    lib = un.plan.lib
    L.check(lib.sr3_plan_invalidate_derived(un.plan.handle))
    with pytest.raises(L.Sr3Error, match='stale'):
        E.unet_forward(un.plan, un.arena.data, un.freq, un._ws, x, noise_level=t)
    other = un.arena.data.clone()                     # same content, another pointer: prepared-from check
    L.check(lib.sr3_plan_prepare_derived(un.plan.handle, L.ptr(other), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    with pytest.raises(L.Sr3Error, match='another arena'):
        E.unet_forward(un.plan, un.arena.data, un.freq, un._ws, x, noise_level=t)
    un._derived_key = None                            # the module path re-prepares from its own arena
    G.assert_close(un(x, t).cpu(), ref, what='after re-prepare')
    # the optimizer-step notification goes through the same door
    un.weights_changed()
    with pytest.raises(L.Sr3Error, match='stale'):
        E.unet_forward(un.plan, un.arena.data, un.freq, un._ws, x, noise_level=t)
    G.assert_close(un(x, t).cpu(), ref, what='after weights_changed')
    # (2) option toggle after a forward
    un.plan.set_option('winograd', 0)
    assert not any(o['tile_cfg'] in (11, 12, 13) for o in un.plan.op_list(x.shape[0]))
    G.assert_close(un(x, t).cpu(), ref, what='winograd off')
    un.plan.set_option('winograd', 1)
    G.assert_close(un(x, t).cpu(), ref, what='winograd on again')
