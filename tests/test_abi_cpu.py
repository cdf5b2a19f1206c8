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


@pytest.mark.parametrize('name', ['sr3_tiny', 'ddpm_tiny', 'sr3_seam', 'sr3_uncond'])
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


@pytest.mark.parametrize('name', ['sr3_tiny', 'ddpm_tiny', 'sr3_seam', 'sr3_uncond'])
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


def test_plan_launch_list_no_gpu():
    """sr3_plan_op_info (host-only): the compiled launch list of the BASELINE config -- which kernel / tile each layer
    gets, what is fused, and that the opt-in split option only re-targets halo-tile convolutions."""
    from sr3_hip import engine as E
    p = E.Plan('sr3', 6, 3, 64, 32, [1, 2, 4, 8, 8], [16], 2, 128)
    # default inference plan: EVERY 3x3 stride-1 conv on the Winograd F(2x2,3x3) kernel (tile 11; round 4: the 8x8 maps too,
    # four images per workgroup tile, split-K); the res_convs of those blocks run as their own 1x1 GEMMs (the Winograd
    # kernel has no second K-segment)
    # plan option fold_fuse (default 1, round 6): 31 of the 61 GroupNorm folds are done by the kernel that completes their (last) source --
    # the split-K reduce of the conv in front of them, or the stand-alone statistics pass -- and leave the launch list
    fops = p.op_list(16)
    assert len(fops) == p.num_ops(16) == 138 and sum(1 for o in fops if o['kind'] == 40) == 30
    assert sum(1 for o in fops if o['kind'] == 30) == 7       # (the 8x8 attention's out conv runs unsplit on the 32-row tile: a statistics pass, which folds)
    assert [o for o in fops if o['kind'] != 40] == [o for o in (p.set_option('fold_fuse', 0), p.op_list(16))[1] if o['kind'] != 40]
    wops = p.op_list(16)                               # (the rest of this test walks the other options with fold_fuse off)
    assert len(wops) == p.num_ops(16) == 169      # (168 + the statistics pass behind the unsplit 8x8 out conv of gemm2; round 3: the input conv writes its own GroupNorm partials: no statistics pass)
    assert wops[1]['kind'] == 20 and wops[1]['fused_output_stats'] and wops[2]['kind'] == 40
    # plan option gemm2 (default 1, round 6): the 1x1 stride-1 convs with Cout % 128 == 0 run the plain GEMM kernel of gemm1x1.hip
    # (tile 22; rows % 64 == 0 and channels % 32 == 0 hold for every layer of this network), reading their weights pre-split in MFMA
    # fragment order from the derived buffer (6 bytes per weight); the Cout = 64 res_convs keep the im2col kernel.  Last session of round 6:
    # Downsample's 3x3 stride-2 convs with Cout % 128 == 0 run the same kernel's stride-2 form (9 k-steps per 32-channel chunk)
    dconvs = [o for o in wops if o['kind'] == 50]
    for o in dconvs:
        s2 = o['ksize'] == 3 and o['stride'] == 2
        # (plan option gemm_n64: Cout % 128 != 0 -- the Cout = 64 layers of the 128 x 128 level -- on the kernel's 64 x 64 tile, waves 2 x 2)
        assert (o['tile_cfg'] == 22) == (((o['ksize'] == 1 and o['stride'] == 1) or s2) and o['cout'] % 64 == 0), o
        if o['tile_cfg'] == 22:        # split-K only below 128 workgroups; tile rows 64, or 32 where 64-row tiles leave slots empty (gemm1x1_rows)
            M = 16 * o['h_out'] * o['w_out']
            cols = 128 if o['cout'] % 128 == 0 else 64
            rows = 64 if (cols == 64 or (M // 64) * (o['cout'] // 128) >= 384) else (32 if (s2 or M <= 1024 or o['cin'] <= 512) else 64)
            # (the stride-2 form: long K on small maps -- it splits whenever its tiles do not fill the 512 workgroup slots)
            assert (o['ksplit'] > 1) == ((M // rows) * (o['cout'] // cols) < (512 if s2 else 128)), o
    assert sum(1 for o in dconvs if o['tile_cfg'] == 22) == 34
    assert [(o['cin'], o['h_out'], o['ksplit']) for o in dconvs if o['tile_cfg'] == 22 and o['ksize'] == 3] == [(64, 64, 1), (128, 32, 1), (256, 16, 2), (512, 8, 4)]
    p.set_option('gemm_n64', 0)        # plan option gemm_n64 = 0: the four Cout = 64 layers back on the im2col kernel (Downsample 64 -> 64 on its fp32 form)
    nops = [o for o in p.op_list(16) if o['kind'] == 50]
    assert [(a['tile_cfg'], b['tile_cfg'], a['cout']) for a, b in zip(dconvs, nops) if a['tile_cfg'] != b['tile_cfg']] == [(22, 2, 64)] + [(22, 16, 64)] * 3
    p.set_option('gemm_n64', 1)
    # plan option fork_side (default 0): every unsplit res_conv is emitted in front of its block's first conv and marked for the side stream,
    # block2's conv -- which adds it as its residual -- waits for it; the embedding MLP beside the input conv, joined by the first FiLM conv.
    # Same ops, same flops; nothing is marked in the default plan
    assert not any('side_id' in o for o in wops)
    p.set_option('fork_side', 1)
    kops = p.op_list(16)
    assert len(kops) == len(wops) and sum(o['flops'] for o in kops) == sum(o['flops'] for o in wops)
    sides = {o['side_id']: i for i, o in enumerate(kops) if o.get('side_id', -1) >= 0}
    waits = {o['wait_id']: i for i, o in enumerate(kops) if o.get('wait_id', -1) >= 0}
    assert sorted(sides) == sorted(waits) == list(range(19))                 # 18 res_convs + the embedding
    for k, i in sides.items():
        a, b = kops[i], kops[waits[k]]
        assert waits[k] > i and b['kind'] == 50 and b['ksize'] == 3
        if a['kind'] == 10:
            assert i == 0 and waits[k] == 3                                    # embedding | input conv, fold -> first block's conv joins
        else:
            assert a['kind'] == 50 and a['ksize'] == 1 and a['ksplit'] == 1 and a['cout'] == b['cout'] == b['cin'] and a['h_out'] == b['h_out']
            assert kops[i + 1]['kind'] == 50 and kops[i + 1]['ksize'] == 3 and kops[i + 1]['cin'] == a['cin']      # block1's conv follows: same input
    p.set_option('fork_side', 0)
    assert p.op_list(16) == wops
    p.set_option('gemm_s2', 0)         # plan option gemm_s2 = 0: the three Downsample convs back on the im2col split tile, nothing else moves
    sops = [o for o in p.op_list(16) if o['kind'] == 50]
    assert [(a['tile_cfg'], b['tile_cfg']) for a, b in zip(dconvs, sops) if a['tile_cfg'] != b['tile_cfg']] == [(22, 2)] + [(22, 16)] * 3
    p.set_option('gemm_s2', 1)
    nbytes_gemm2 = int(p.lib.sr3_plan_derived_bytes(p.handle))
    p.set_option('gemm2', 0)           # the rest of this test walks the im2col options with gemm2 off
    assert nbytes_gemm2 - int(p.lib.sr3_plan_derived_bytes(p.handle)) == 6 * sum(o['cout'] * o['cin'] * o['ksize'] ** 2 for o in dconvs if o['tile_cfg'] == 22)
    wops0 = p.op_list(16)
    assert len(wops0) == p.num_ops(16) == 168
    for a, b in zip([o for o in wops if o['kind'] == 50], [o for o in wops0 if o['kind'] == 50]):
        assert a['flops'] == b['flops'] and (a['tile_cfg'] == b['tile_cfg'] or (a['tile_cfg'] == 22 and b['tile_cfg'] in (16, 2)))     # (2: Downsample 64 -> 64 on the fp32 im2col tile)
    wops = wops0
    wconvs = [o for o in wops if o['kind'] == 50]
    for o in wconvs:
        assert (o['tile_cfg'] in (11, 12, 13)) == (o['ksize'] == 3 and o['stride'] == 1), o
        # default plan options wino_split = 1, wino2 = 1 (round 6): maps >= 16x16 on the 3 x bf16 split arithmetic as two four-wave
        # workgroups per CU (conv3x3_wino2.hip, reported as tile 13), and (wino_split8, round 5) the four-image tile of the 8x8 maps on
        # the 8-wave kernel's split instantiation (tile 12)
        assert (o['tile_cfg'] in (12, 13)) == (o['ksize'] == 3 and o['stride'] == 1), o
        assert (o['tile_cfg'] == 13) == (o['ksize'] == 3 and o['stride'] == 1 and o['h_out'] >= 16), o
        assert not o['fused_res_conv_cin']
        if o['tile_cfg'] in (11, 12) and o['h_out'] == 8:       # the four-image tile has no direct epilogue: always split-K, at most
            assert o['ksplit'] >= 2 and -(-o['cin'] // 16) <= 16 * o['ksplit'], o      # 16 chunks (256 channels) per split
    # the im2col SPLIT tiles split their weights while staging them by default (tiles 14-17); plan option gemm_wpre = 1 (round 6's form,
    # measured slower again, kept as an A/B knob): they read them pre-split and in MFMA fragment order straight from the derived buffer
    # (tiles 18-21), three bf16 planes = 6 bytes per weight of the res_convs, attention projections and Downsample convs with Cout > 64
    assert all(not 18 <= o['tile_cfg'] <= 21 for o in wconvs) and any(14 <= o['tile_cfg'] <= 17 for o in wconvs)
    nbytes_nopre = int(p.lib.sr3_plan_derived_bytes(p.handle))
    p.set_option('gemm_wpre', 1)
    pre = [o for o in p.op_list(16) if o['kind'] == 50]
    nbytes_both = int(p.lib.sr3_plan_derived_bytes(p.handle))
    nbytes_wsplit = nbytes_both - nbytes_nopre
    n_w = sum(o['cout'] * o['cin'] * o['ksize'] ** 2 for o in pre if 18 <= o['tile_cfg'] <= 21)
    assert n_w > 0 and nbytes_wsplit == 6 * n_w           # (every such layer of this network has Cout, Cin multiples of 32: no padding)
    assert all((18 <= a['tile_cfg'] <= 21) == (14 <= b['tile_cfg'] <= 17) and a['tile_cfg'] in (b['tile_cfg'], b['tile_cfg'] + 4)
               for a, b in zip(pre, wconvs))
    p.set_option('gemm_split', 0)
    assert int(p.lib.sr3_plan_derived_bytes(p.handle)) == nbytes_nopre      # nothing pre-split without the split tiles
    p.set_option('gemm_split', 1)
    assert int(p.lib.sr3_plan_derived_bytes(p.handle)) == nbytes_both
    p.set_option('gemm_wpre', 0)
    assert p.op_list(16) == wops
    nbytes_both, nbytes_wsplit = nbytes_nopre, 0
    p.set_option('wino_split', 0)                         # the exact-fp32 MFMA instantiation everywhere: same list, tile 11
    eops = p.op_list(16)
    assert len(eops) == len(wops)
    for a, b in zip(wops, eops):
        assert b['tile_cfg'] == (11 if a['tile_cfg'] in (12, 13) else a['tile_cfg']) and a['flops'] == b['flops']
        assert a['ksplit'] == b['ksplit'] or a['tile_cfg'] == 13      # (the 8 x 16 tile fills 512 workgroup slots: its own split-K choice)
    # the derived buffer holds both forms of every filter under wino_split (fp32 + 1.5x that for the three bf16 planes)
    assert abs((nbytes_both - nbytes_wsplit) / (int(p.lib.sr3_plan_derived_bytes(p.handle)) - nbytes_wsplit) - 2.5) < 1e-6
    p.set_option('wino_split', 1)
    p.set_option('wino2', 0)                              # the 8-wave kernel everywhere (round 5's plan): tile 12
    for a, b in zip(wops, p.op_list(16)):
        assert b['tile_cfg'] == (12 if a['tile_cfg'] == 13 else a['tile_cfg']) and a['flops'] == b['flops']
    p.set_option('wino2', 1)
    assert p.op_list(16) == wops
    # a batch that is not a multiple of 4 keeps the direct halo kernel on the 8x8 maps
    for o in p.op_list(3):
        if o['kind'] == 50 and o['ksize'] == 3 and o['stride'] == 1:
            assert (o['tile_cfg'] == 13) == (o['h_out'] >= 16) and o['tile_cfg'] not in (11, 12), o
    assert sum(1 for o in wconvs if o['ksize'] == 1) == 12 + 18
    assert abs(sum(o['flops'] for o in wops) / 16 / 1e9 - 92.18) < 0.05       # algorithmic FLOPs do not change
    assert int(p.lib.sr3_plan_derived_bytes(p.handle)) > 0
    # the direct kernels (plan option winograd = 0; also what the training plan and an explicit tile_cfg use)
    p.set_option('winograd', 0)
    ops = p.op_list(16)
    assert len(ops) == p.num_ops(16) == 150 + 11
    convs = [o for o in ops if o['kind'] == 50]
    assert sum(1 for o in ops if o['kind'] == 60) == 6 and sum(1 for o in ops if o['kind'] == 40) == 61
    # every 3x3 stride-1 conv runs on the halo-tile kernel; 1x1 and stride-2 convs on the im2col kernel
    for o in convs:
        halo = 5 <= o['tile_cfg'] <= 10
        assert halo == (o['ksize'] == 3 and o['stride'] == 1), o
        # 1x1 / stride-2 convs: the im2col kernel's 64x64 tile on its 3 x bf16 split instantiation (plan option gemm_split,
        # default 1; reported as tile 16 = the split form of tile 3; 20 with pre-split weights under gemm_wpre = 1); the 9-tap
        # Downsample with Cout <= 64 stays on the fp32 MFMA
        if not halo:
            assert o['tile_cfg'] == (2 if (o['ksize'] == 3 and o['cout'] <= 64) else 16), o
        if o['fused_res_conv_cin']:
            assert halo and not o['upsample']
    # 18 ResnetBlocks change their channel count: where block2's conv runs unsplit (the 128x128 and 64x64 levels) their
    # 1x1 res_conv rides inside its launch; under split-K (32x32 and below at batch 16) it stays a 1x1 GEMM of its own
    assert sum(1 for o in convs if o['fused_res_conv_cin']) == 7
    assert all(o['ksplit'] == 1 for o in convs if o['fused_res_conv_cin'])
    assert sum(1 for o in convs if o['ksize'] == 1) == 12 + 11         # qkv + out of the 6 attention blocks + 11 res_convs
    # Cout <= 64 layers: 256x64 tile; Cout > 64 layers whose 256x128 tiling still gives one workgroup per CU (256): the
    # 8-wave tile; the remaining small-M layers: 128x128 + split-K
    for o in convs:
        if 5 <= o['tile_cfg'] <= 10:
            wg9 = 16 * (o['h_out'] // 16) * (o['w_out'] // 16) * -(-o['cout'] // 128) if o['h_out'] >= 16 else 0
            if o['cout'] <= 64:
                assert o['tile_cfg'] == 6
            elif wg9 >= 256:
                assert o['tile_cfg'] == 9 and o['ksplit'] == 1, o
            else:
                assert o['tile_cfg'] == 5 and o['ksplit'] > 1, o
    total = sum(o['flops'] for o in ops) / 16 / 1e9
    assert abs(total - 92.18) < 0.05 and abs(p.forward_flops(16) / 16 / 1e9 - 92.35) < 0.05
    # gemm_split = 0: the same list with those convs on the exact-fp32 MFMA (1x1: the 64x64 tile)
    p.set_option('gemm_split', 0)
    for a, b in zip(ops, p.op_list(16)):
        assert a['kind'] == b['kind'] and a['flops'] == b['flops']
        if a['kind'] == 50 and 14 <= a['tile_cfg'] <= 17:
            assert b['tile_cfg'] == a['tile_cfg'] - 13 == 3 and a['ksplit'] == b['ksplit'], (a, b)
        else:
            assert a['tile_cfg'] == b['tile_cfg'] and a['ksplit'] == b['ksplit']
    p.set_option('gemm_split', 1)
    assert p.op_list(16) == ops
    from helpers import experiments_built
    from sr3_hip import lib as L
    if experiments_built():
        # opt-in split mode: same list, only the halo tiles change (5 -> 7 or 10, 6 -> 8, 9 -> 10)
        p.set_option('split_bf16', 1)
        ops2 = p.op_list(16)
        assert len(ops2) == len(ops)
        for a, b in zip(ops, ops2):
            assert a['kind'] == b['kind'] and a['flops'] == b['flops']
            if a['kind'] == 50 and 5 <= a['tile_cfg'] <= 10:
                assert b['tile_cfg'] in {5: (7, 10), 6: (8,), 9: (10,)}[a['tile_cfg']], (a, b)
            else:
                assert a['tile_cfg'] == b['tile_cfg'] and a['ksplit'] == b['ksplit']
        p.set_option('split_bf16', 0)
        assert p.op_list(16) == ops
    else:        # the default build refuses the options that select experiment kernels
        with pytest.raises(L.Sr3Error) as ei:
            p.set_option('split_bf16', 1)
        assert 'SR3_EXPERIMENTS' in str(ei.value)
        assert p.op_list(16) == ops
    # batch 1: everything is small-M
    assert all(o['tile_cfg'] != 9 or o['h_out'] >= 128 for o in p.op_list(1) if o['kind'] == 50)


def test_missing_library_fails_loudly():
    """No fallback: when the shared library is absent the product path raises with a build hint (checked in a
    subprocess through the SR3_LIBRARY override so this process keeps its loaded library)."""
    import subprocess
    import sys
    pkg = os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd')
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from sr3_hip import lib as L\n"
            "try:\n    L.load()\nexcept L.Sr3Error as e:\n    print('RAISED', e)\n" % pkg)
    env = dict(os.environ, SR3_LIBRARY='/nonexistent/libsr3_mi355x.so')
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120).stdout
    assert 'RAISED' in out and 'not built' in out and 'no CPU / eager fallback' in out


def test_plan_options_are_validated():
    from sr3_hip import engine as E, lib as L
    p = E.Plan('sr3', 6, 3, 8, 4, [1, 2], [8], 1, 16)
    with pytest.raises(L.Sr3Error):
        p.set_option('no_such_option', 1)
    assert p.set_option('loss_l2', 1) == 0 and p.set_option('loss_l2', 0) == 1          # returns the previous value
    n0 = p.num_ops(2)
    p.set_option('fuse_stats', 0)
    assert p.num_ops(2) > n0                       # stand-alone statistics passes come back
    p.set_option('fuse_stats', 1)
    assert p.num_ops(2) == n0 and p.workspace_bytes(2) > 0
    with pytest.raises(L.Sr3Error):
        p.workspace_bytes(0)


def test_dropout_threshold_matches_the_oracle_mask():
    """The engine derives its keep threshold from the decimal p the fp32 ABI argument stands for, so its mask is the
    oracle's (int(p * 2^32) on the Python float) bit for bit -- a threshold off by a few counts flips a handful of mask
    elements in a 64-image batch and shows up as a 1e-4 relative error in the conv weight gradients of those blocks."""
    import ctypes as C
    import numpy as np
    from sr3_hip import lib as L
    lib = L.load()
    for p in (0.2, 0.1, 0.5, 0.25, 0.05, 0.3, 0.123):
        s = C.c_float()
        t = lib.sr3_dropout_threshold(C.c_float(p), C.byref(s))
        assert t == int(p * 4294967296.0), (p, t)
        assert np.float32(s.value) == np.float32(1.0 / (1.0 - p)), (p, s.value)
    assert lib.sr3_dropout_threshold(C.c_float(0.0), None) == 0


def test_round5_plan_options_no_gpu():
    """The plan options added in round 5 are accepted, return the previous value, and the structural one changes the launch
    list as documented (include/sr3_mi355x.h): wino_split8 moves the four-image 8x8 tile between tiles 12 and 11; attn_split /
    wgrad_split are run-time switches that leave the list alone; gemm_wpre moves the im2col split tiles to 18-21."""
    from sr3_hip import engine as E
    p = E.Plan('sr3', 6, 3, 64, 32, [1, 2, 4, 8, 8], [16], 2, 128)
    ops = p.op_list(16)
    w8 = [o for o in ops if o['kind'] == 50 and o['ksize'] == 3 and o['stride'] == 1 and o['h_out'] == 8]
    assert len(w8) == 14 and all(o['tile_cfg'] == 12 and o['ksplit'] >= 2 for o in w8)
    assert p.set_option('wino_split8', 0) == 1
    ops0 = p.op_list(16)
    assert [o['tile_cfg'] for o in ops0 if o['kind'] == 50 and o['ksize'] == 3 and o['stride'] == 1 and o['h_out'] == 8] == [11] * 14
    assert all(a['tile_cfg'] == b['tile_cfg'] for a, b in zip(ops, ops0) if not (a['kind'] == 50 and a['h_out'] == 8 and a['ksize'] == 3))
    assert p.set_option('wino_split8', 1) == 0 and p.op_list(16) == ops
    for key in ('attn_split', 'wgrad_split'):
        assert p.set_option(key, 0) == 1 and p.op_list(16) == ops
        assert p.set_option(key, 1) == 0
    # gemm_wpre (default 0: the weights pre-split in MFMA fragment order, read straight from global memory, measured slower): 16 <-> 20
    # (since gemm_s2 / gemm_n64 every 1x1 and stride-2 conv of this network runs the plain GEMM kernel, tile 22: the im2col split tiles
    # appear with gemm2 = 0)
    assert sorted(set(o['tile_cfg'] for o in ops if o['kind'] == 50 and 14 <= o['tile_cfg'] <= 21)) == []
    assert p.set_option('gemm2', 0) == 1
    assert sorted(set(o['tile_cfg'] for o in p.op_list(16) if o['kind'] == 50 and 14 <= o['tile_cfg'] <= 21)) == [16]
    assert p.set_option('gemm_wpre', 1) == 0
    assert sorted(set(o['tile_cfg'] for o in p.op_list(16) if o['kind'] == 50 and 14 <= o['tile_cfg'] <= 21)) == [20]
    assert p.set_option('gemm_wpre', 0) == 1 and p.set_option('gemm2', 1) == 0 and p.op_list(16) == ops
    # wino2 (round 6): maps >= 16 x 16 between the two-workgroups-per-CU kernel (tile 13) and the 8-wave kernel (tile 12)
    assert any(o['tile_cfg'] == 13 for o in ops) and p.set_option('wino2', 0) == 1
    assert not any(o['tile_cfg'] == 13 for o in p.op_list(16)) and p.set_option('wino2', 1) == 0 and p.op_list(16) == ops
    with pytest.raises(Exception):
        p.set_option('no_such_option', 1)
