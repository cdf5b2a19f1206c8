#!/usr/bin/env python3
"""bench.py -- sampling throughput of the MI355X engine on the BASELINE.json configurations.

Default workload (BASELINE.json configs[1], the one the headline metric is quoted on): the SR3 16->128 UNet of
config/sr_sr3_16_128.json (inner 64, mults 1,2,4,8,8, attention at 16x16, 97.8 M fp32 parameters, random init),
batch 16 per GPU, T = 2000 linear-beta reverse steps.  A *step* is one reverse step p_sample of the whole batch:
[z ~ N(0,I)] -> UNet forward -> fused x_{t-1} update -> counter decrement, replayed from one hipGraph.  The default
--steps 2000 times one complete sample; images/s = N * B / (T * t_step).  `--config sr3_64_512` (configs[3], batch 4)
and `--config ddpm_128` (configs[4], batch 32, unconditional) time the other BASELINE.json networks the same way.

One JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline            -- dominant kernel (a halo-tile 3x3 conv instantiation on v_mfma_f32_32x32x2_f32): algorithmic
                         FLOPs per launch / average launch duration, measured with HIP events around every launch of
                         the plan (sr3_unet_forward_profile, on the stream the kernels run on) after the timed region.
  parity              -- the graph bench.py just timed, replayed once from a fixed (x, cond, t); its eps and x_{t-1}
                         against the CPU oracle on the same inputs and the same in-graph z (max abs difference).
  cpu_baseline        -- the reference itself (subprocess importing $SR3_REFERENCE or /root/reference, kind
                         "reference") or, where that tree does not exist (the GPU box), the CPU oracle (kind "port"),
                         timed on this node's host cores on a bounded sample; `.train` = training steps per image.
  torch_rocm_baseline -- the same oracle ops on `cuda` through stock PyTorch-ROCm (MIOpen / rocBLAS): "what you get by
                         default on this node".  Reported, never the target.
  train               -- BASELINE.json configs[2] / [4]: training step (p_losses + backward + Adam) images/s.
Multi-GPU: one process per GPU.  `python bench.py --gpus N` launches the N ranks itself (torch.distributed.run over
127.0.0.1) when it is not already running under a launcher; under `python -m torch.distributed.run ... bench.py --gpus N`
it uses the launcher's ranks (and refuses a WORLD_SIZE that differs from --gpus).  Sampling: independent image batches
per rank (the reverse chains share nothing), no collective in the data path; training: RCCL all-reduce of gradient
buckets overlapped with the backward.  The timed region is bracketed by barrier + synchronize and the MAX over ranks is
reported ("scaling": "weak").
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd')

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (~2.5 PF)
PROFILE_ROUND = 'r06'
# sr3_unet_forward_profile op kinds (plan.hip): Winograd fp32 one-image / four-image tile, SPLIT one-image / four-image tile (8-wave
# kernel), 575 = the two-workgroups-per-CU SPLIT kernel of conv3x3_wino2.hip; im2col SPLIT tiles 14-17
WINO_KINDS = (455, 465, 555, 565, 575)
WINO_SPLIT_KINDS = (555, 565, 575)
IGEMM_SPLIT_KINDS = (651, 652, 653, 654, 232)         # (232: the plain 1x1 GEMM kernel of gemm1x1.hip, plan option gemm2; round 6)              # profiles/<round>_hbm_traffic.json, <round>_sq_counters.json feed `roofline`

# The `model` subtrees of the reference's configs (config/sr_sr3_16_128.json:39-77, sr_sr3_64_512.json:39-80,
# sample_ddpm_128.json:38-79) + the batch sizes BASELINE.json quotes.
CONFIGS = {
    'sr3_16_128': dict(which='sr3', unet=dict(in_channel=6, out_channel=3, inner_channel=64, channel_multiplier=[1, 2, 4, 8, 8],
                                              attn_res=[16], res_blocks=2, dropout=0.2),
                       beta=(1e-6, 1e-2), size=128, conditional=True, batch=16, train_batch=64, lr=1e-4,
                       baseline_cfg=1, title='SR3 16->128', ref_json='config/sr_sr3_16_128.json'),
    'sr3_64_512': dict(which='sr3', unet=dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=16,
                                              channel_multiplier=[1, 2, 4, 8, 16], attn_res=[], res_blocks=1, dropout=0),
                       beta=(1e-6, 1e-2), size=512, conditional=True, batch=4, train_batch=2, lr=3e-6,
                       baseline_cfg=3, title='SR3 64->512', ref_json='config/sr_sr3_64_512.json'),
    'ddpm_128': dict(which='ddpm', unet=dict(in_channel=3, out_channel=3, inner_channel=64, channel_multiplier=[1, 1, 2, 2, 4, 4],
                                             attn_res=[16], res_blocks=2, dropout=0.2),
                     beta=(1e-4, 2e-2), size=128, conditional=False, batch=32, train_batch=32, lr=1e-4,
                     baseline_cfg=4, title='DDPM 128x128 (unconditional)', ref_json='config/sample_ddpm_128.json'),
}


def config_opt(name, n_timestep=2000, phase='val'):
    c = CONFIGS[name]
    sched = dict(schedule='linear', n_timestep=n_timestep, linear_start=c['beta'][0], linear_end=c['beta'][1])
    return {
        'phase': phase, 'gpu_ids': [0], 'distributed': False,
        'path': {'checkpoint': '/tmp', 'resume_state': None},
        'train': {'optimizer': {'type': 'adam', 'lr': c['lr']}},
        'model': {
            'which_model_G': c['which'], 'finetune_norm': False, 'unet': dict(c['unet']),
            'beta_schedule': {'train': dict(sched), 'val': dict(sched)},
            'diffusion': dict(image_size=c['size'], channels=3, conditional=c['conditional']),
        },
    }


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))      # torch's intra-op pool stops scaling (and thrashes) far below 256 threads


def reference_root():
    for p in (os.environ.get('SR3_REFERENCE'), '/root/reference'):
        if p and os.path.isfile(os.path.join(p, 'model', 'networks.py')) and os.path.realpath(p) != os.path.realpath(PKG):
            return p
    return None


# ---------------------------------------------------------------------------------------------------------------
# parity of the timed graph + CPU / stock-PyTorch baselines
# ---------------------------------------------------------------------------------------------------------------
def capture_parity_inputs(netG, st, cfg, T):
    """Replay the graph that was just timed ONCE from a fixed (x, cond, t) and keep what the oracle needs to redo
    that step on the CPU: inputs, the z the graph drew, the graph's eps and x_{t-1}."""
    import torch
    dev = st['img'].device
    t = T // 2 + 7
    g = torch.Generator().manual_seed(4242)
    x = torch.randn(tuple(st['img'].shape), generator=g)
    cond = (torch.rand(tuple(st['img'].shape), generator=g) * 2 - 1) if st['cond'] is not None else None
    st['img'].copy_(x)
    if cond is not None:
        st['cond'].copy_(cond)
    st['step'].fill_(t)
    st['graph'].replay()
    torch.cuda.synchronize(dev)
    return dict(t=t, x=x, cond=cond, z=st['z'].cpu(), eps=st['eps'].cpu(), x_next=st['img'].cpu())


def oracle_tools(cfg_name, netG):
    import torch
    from oracle import sr3_oracle as O
    opt = config_opt(cfg_name)
    desc = O.desc_from_opt(opt)
    tab = O.schedule_tables(opt['model']['beta_schedule']['val'])
    sd = {k: v.detach().cpu().clone() for k, v in netG.state_dict().items()}
    return O, desc, tab, sd, torch


def parity_vs_oracle(cfg_name, netG, par):
    """max |GPU - oracle| of the replayed step (same x, cond, t, z, weights).  Returns (record, oracle seconds, x_ref)."""
    O, desc, tab, sd, torch = oracle_tools(cfg_name, netG)
    x, t, B = par['x'], par['t'], par['x'].shape[0]
    t0 = time.time()
    with torch.no_grad():
        # O.p_sample (sr3 diffusion.py:151-174) unrolled so that eps is kept as well
        if desc['variant'] == 'sr3':
            level = torch.FloatTensor([tab['sqrt_alphas_cumprod_prev'][t + 1]]).repeat(B, 1)
        else:
            level = torch.full((B,), t, dtype=torch.long)
        inp = torch.cat([par['cond'], x], dim=1) if par['cond'] is not None else x
        eps_ref = O.unet_forward(sd, desc, inp, level)
        x_ref = O.p_sample_update(tab, x, eps_ref, t, par['z'])
    dt = time.time() - t0
    err_x = float((par['x_next'] - x_ref).abs().max())
    err_e = float((par['eps'] - eps_ref).abs().max())
    tol_x = 2e-5 * max(1.0, float(x_ref.abs().max()))
    tol_e = 2e-5 * max(1.0, float(eps_ref.abs().max()))
    rec = dict(parity_max_abs=max(err_x, err_e), x_next_max_abs_err=err_x, eps_max_abs_err=err_e,
               what='one hipGraph-replayed reverse step (the graph that was timed) at t=%d, batch %d: eps and x_{t-1} vs the CPU '
                    'oracle on the same x, cond, in-graph z and weights' % (t, B),
               eps_ref_max_abs=float(eps_ref.abs().max()), x_ref_max_abs=float(x_ref.abs().max()),
               tolerance='2e-5 * max(1, |ref|_inf)', ok=bool(err_x <= tol_x and err_e <= tol_e))
    return rec, dt, x_ref


def cpu_baseline(cfg_name, netG, par, budget_s=25.0, train_batch=4):
    """Sampling: p_sample steps at the config's batch (the first one doubles as the parity check); training: Adam
    steps at a reduced batch, per image.  Reference code in a subprocess when its tree is present, else the oracle."""
    import torch
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    B = par['x'].shape[0]
    out = {}
    ref = reference_root()
    parity, t_first, x_ref = parity_vs_oracle(cfg_name, netG, par)       # also the oracle's warm-up at the timed batch
    out['_parity'] = parity
    if ref is not None:
        try:
            out.update(reference_cpu_baseline(ref, cfg_name, netG, par, ncores, budget_s, train_batch))
            return out
        except Exception as e:                                # fall through to the port, but say why
            out['reference_error'] = '%s: %s' % (type(e).__name__, e)
    O, desc, tab, sd, _ = oracle_tools(cfg_name, netG)
    times = []
    x = x_ref
    with torch.no_grad():
        n = 0
        start = time.time()
        while n < 8 and (n == 0 or (time.time() - start) * (n + 1) / n < budget_s):
            t1 = time.time()
            x = O.p_sample(sd, desc, tab, x, par['t'] - 1 - n, par['z'], condition_x=par['cond'])
            times.append(time.time() - t1)
            n += 1
    t_step = sum(times) / len(times)
    out.update(value=B / (2000.0 * t_step), unit='images/s', cores=int(torch.get_num_threads()), kind='port',
               sample='%d reverse steps (oracle p_sample: UNet forward + update) at batch %d after 1 warm-up step at the '
                      'same batch (%.2f s), %.2f s/step, extrapolated x2000' % (len(times), B, t_first, t_step))
    if CONFIGS[cfg_name]['size'] <= 128:          # a 512^2 Adam step on the host is minutes: the bounded sample skips it
        try:
            out['train'] = oracle_train_baseline(cfg_name, sd, min(train_batch, CONFIGS[cfg_name]['train_batch']), 'cpu', ncores)
        except Exception as e:
            out['train'] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


def oracle_train_baseline(cfg_name, sd, batch, device, cores, steps=3):
    """1 warm-up + `steps` timed optimize_parameters-equivalents (model/model.py:48-58: zero_grad -> p_losses -> /numel ->
    backward -> Adam) with torch autograd over the oracle's functional ops on `device`."""
    import torch
    from oracle import sr3_oracle as O
    c = CONFIGS[cfg_name]
    opt = config_opt(cfg_name, phase='train')
    desc = O.desc_from_opt(opt)
    tab = O.schedule_tables(opt['model']['beta_schedule']['train'])
    params = {k: v.to(device).clone().requires_grad_(v.is_floating_point() and k.startswith('denoise_fn.') and 'inv_freq' not in k)
              for k, v in sd.items()}
    optim = torch.optim.Adam([p for p in params.values() if p.requires_grad], lr=c['lr'])
    S = c['size']
    g = torch.Generator().manual_seed(9)
    hr = (torch.rand(batch, 3, S, S, generator=g) * 2 - 1).to(device)
    sr = (torch.rand(batch, 3, S, S, generator=g) * 2 - 1).to(device)
    times = []
    for it in range(steps + 1):
        if device != 'cpu':
            torch.cuda.synchronize()
        t0 = time.time()
        optim.zero_grad()
        z = torch.randn_like(hr)
        if c['which'] == 'sr3':
            gamma = torch.rand(batch, device=device) * 0.5 + 0.4
            loss = O.p_losses_sr3(params, desc, hr, sr, gamma, z, conditional=True)
        else:
            t = torch.randint(0, 2000, (batch,), device=device)
            loss = O.p_losses_ddpm(params, desc, tab, hr, sr, t, z, conditional=False)
        (loss / hr.numel()).backward()
        optim.step()
        float(loss)                                           # the reference's .item() (model/model.py:58)
        if device != 'cpu':
            torch.cuda.synchronize()
        if it > 0:
            times.append(time.time() - t0)
    t_step = sum(times) / len(times)
    return dict(value=batch / t_step, unit='images/s', s_per_step=t_step, batch=batch, cores=cores,
                sample='%d Adam steps at batch %d after 1 warm-up (autograd over the oracle ops on %s; the dropout of the '
                       'training config is left out of this baseline: it flatters the baseline -- the reference itself, dropout '
                       'included, takes 1.76x as long per step on the same cores, profiles/r04_ref_vs_port_cpu.json)'
                       % (len(times), batch, device))


def reference_cpu_baseline(ref, cfg_name, netG, par, cores, budget_s, train_batch):
    """tools/ref_baseline.py in a subprocess: the reference's own `model` package collides with the drop-in's name."""
    import torch
    tmp = tempfile.mkdtemp(prefix='sr3_ref_')
    state = os.path.join(tmp, 'state.pth')
    torch.save({'sd': {k: v.detach().cpu() for k, v in netG.state_dict().items()}, 'x': par['x'], 'cond': par['cond'],
                't': par['t']}, state)
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'ref_baseline.py'), '--ref', ref, '--config', cfg_name, '--state', state,
           '--threads', str(cores), '--budget', str(budget_s), '--train-batch', str(train_batch)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=budget_s * 6 + 120)
    if r.returncode != 0:
        raise RuntimeError('ref_baseline.py rc %d: %s' % (r.returncode, r.stderr.decode()[-400:]))
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


def torch_rocm_baseline(cfg_name, netG, par, dev, steps=5, train_batch=None):
    """The same oracle ops on the GPU through stock PyTorch-ROCm (MIOpen convolutions, rocBLAS GEMMs, eager): the
    'unmodified reference on this node' number for sampling and training.  Also a parity cross-check of the engine."""
    O, desc, tab, sd, torch = oracle_tools(cfg_name, netG)
    B = par['x'].shape[0]
    if train_batch is None:
        train_batch = CONFIGS[cfg_name]['train_batch']      # the batch the `train` leg is quoted on
    sdd = {k: v.to(dev) for k, v in sd.items()}
    x, z = par['x'].to(dev), par['z'].to(dev)
    cond = None if par['cond'] is None else par['cond'].to(dev)
    out = {}
    with torch.no_grad():
        t0 = time.time()
        x1 = O.p_sample(sdd, desc, tab, x, par['t'], z, condition_x=cond)     # warm-up: MIOpen find / kernel build
        torch.cuda.synchronize(dev)
        warm = time.time() - t0
        out['max_abs_diff_vs_engine'] = float((x1.cpu() - par['x_next']).abs().max())
        t0 = time.time()
        for i in range(steps):
            x1 = O.p_sample(sdd, desc, tab, x1, par['t'] - 1 - i, z, condition_x=cond)
        torch.cuda.synchronize(dev)
        t_step = (time.time() - t0) / steps
    out.update(value=B / (2000.0 * t_step), unit='images/s', ms_per_step=t_step * 1e3, kind='oracle ops on cuda (stock MIOpen / rocBLAS, eager)',
               sample='%d reverse steps at batch %d after 1 warm-up step (%.1f s incl. MIOpen find), extrapolated x2000'
                      % (steps, B, warm))
    del sdd, x1
    torch.cuda.empty_cache()
    try:
        out['train'] = oracle_train_baseline(cfg_name, sd, train_batch, str(dev), 0)
        out['train'].pop('cores', None)
    except Exception as e:
        out['train'] = {'error': '%s: %s' % (type(e).__name__, e)}
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------------
# roofline of the dominant kernel
# ---------------------------------------------------------------------------------------------------------------
def roofline_from_profile(netG, x, cond, reps=3):
    """HIP-event timing of every launch of one forward; aggregates the dominant kernel."""
    import torch
    from sr3_hip import lib as L
    un = netG.denoise_fn
    plan = un.plan
    lib = L.load()
    B = x.shape[0]
    if plan.variant == 'sr3':
        level, tstep = torch.full((B,), 0.5, device=x.device), None
    else:
        level, tstep = None, torch.full((B,), 1000, dtype=torch.long, device=x.device)
    wsbuf, need = un._ws.get(plan, B, x.device)
    out = torch.empty(B, 3, plan.image_size, plan.image_size, device=x.device)
    max_ops = 4096
    ms = (C.c_float * max_ops)()
    kind = (C.c_int * max_ops)()
    fl = (C.c_double * max_ops)()
    n = C.c_int()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    agg = {}
    for r in range(reps + 1):
        L.check(lib.sr3_unet_forward_profile(plan.handle, L.ptr(x), L.ptr(cond), 0 if cond is None else cond.shape[1],
                                             L.ptr(level), L.ptr(tstep), L.ptr(un.freq), L.ptr(un.arena.data), L.ptr(wsbuf),
                                             need, L.ptr(out), B, stream, max_ops, ms, kind, fl, C.byref(n)))
        if r == 0:
            continue      # warm-up
        for i in range(n.value):
            a = agg.setdefault(kind[i], [0.0, 0.0, 0])
            a[0] += ms[i]
            a[1] += fl[i]
            a[2] += 1
    # template arguments: <WAVES_M, WAVES_N, fused 1x1 segment, dropout, 0 = fp32 MFMA | 1 = 3 x bf16 split MFMA,
    #                      32x32 MFMA tiles across a wave's tile>
    names = {55: 'k_conv3x3_halo<2,2,false,false,0,2>', 56: 'k_conv3x3_halo<4,1,false,false,0,2>',
             57: 'k_conv3x3_halo<2,2,true,false,0,2>', 58: 'k_conv3x3_halo<4,1,true,false,0,2>',
             255: 'k_conv3x3_halo<4,2,false,false,0,2>', 257: 'k_conv3x3_halo<4,2,true,false,0,2>',
             155: 'k_conv3x3_halo<2,2,false,false,1,2>', 157: 'k_conv3x3_halo<2,2,true,false,1,2>',
             156: 'k_conv3x3_halo<4,2,false,false,1,1>', 158: 'k_conv3x3_halo<4,2,true,false,1,1>',
             355: 'k_conv3x3_halo<4,2,false,false,1,2>', 357: 'k_conv3x3_halo<4,2,true,false,1,2>',
             455: 'k_conv3x3_wino<0,false,false,false>', 465: 'k_conv3x3_wino<0,false,true,false>',
             555: 'k_conv3x3_wino<0,false,false,true>', 565: 'k_conv3x3_wino<0,false,true,true>',
             575: 'k_conv3x3_wino2<0>'}
    total_ms = sum(a[0] for a in agg.values()) / reps
    dom = max((k for k in names if k in agg), key=lambda k: agg[k][0])     # largest share of the forward
    t_ms, flops, launches = agg[dom]
    achieved = flops / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0
    halo_all = [agg[k] for k in names if k in agg]
    all_tf = sum(a[1] for a in halo_all) / (sum(a[0] for a in halo_all) * 1e-3) / 1e12
    detail = {str(k): dict(ms_per_forward=v[0] / reps, launches_per_forward=v[2] // reps,
                           tflops=(v[1] / (v[0] * 1e-3) / 1e12 if v[0] > 0 and v[1] > 0 else None))
              for k, v in sorted(agg.items())}
    traffic = None
    counters = None
    kname = 'sr3::' + names[dom].replace(',', ', ')

    def by_kernel(table):
        # the committed summaries are keyed by the exact kernel symbol of the tree they were recorded on (tools/round_profile.sh
        # after the last kernel change of the round): no match, no number
        return table.get(kname)
    try:        # HBM bytes per launch from the committed rocprofv3 PMC passes of this command (profiles/)
        with open(os.path.join(ROOT, 'profiles', PROFILE_ROUND + '_hbm_traffic.json')) as f:
            traffic = by_kernel(json.load(f))['hbm_bytes_per_launch']
    except (OSError, KeyError, TypeError, ValueError):
        pass
    try:        # SQ counters of the current kernels (MFMA-busy fraction), same provenance
        with open(os.path.join(ROOT, 'profiles', PROFILE_ROUND + '_sq_counters.json')) as f:
            sq = json.load(f)
        counters = {'dominant_kernel': by_kernel(sq), 'attention': sq.get('attention')}
    except (OSError, ValueError):
        pass
    rocprof_avg_us = None
    try:        # the same kernel's average duration in the committed rocprofv3 --kernel-trace --stats summary of this command
        import csv
        for row in csv.DictReader(open(os.path.join(ROOT, 'profiles', PROFILE_ROUND + '_bench_kernel_stats.csv'))):
            if row['Name'].replace('void ', '').startswith(kname + '('):
                rocprof_avg_us = float(row['AverageNs']) / 1e3
                break
    except (OSError, KeyError, ValueError):
        pass
    is_wino = dom in WINO_KINDS
    is_split = dom in WINO_SPLIT_KINDS or ((not is_wino) and names[dom].split(',')[4] == '1')
    # split kernels: six bf16 MFMA products per fp32 product -> fp32-equivalent peak = bf16 dense peak / 6
    peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if is_split else FP32_MFMA_PEAK_TFLOPS
    # Winograd F(2x2,3x3): 16 multiplies per 2x2 output block and (cin, cout) pair instead of 36, so the MFMA pipe executes
    # 1 / 2.25 of the direct-convolution FLOPs SURVEY.md 8d counts.  `achieved` / `frac` are the EXECUTED MFMA rate against the
    # fp32 MFMA roof (a fraction of a roof, <= 1); the direct-convolution-equivalent rate is reported beside it.
    executed = achieved / 2.25 if is_wino else achieved
    # step level: the floor the design chose = (Winograd FLOPs / 2.25 + all other contraction FLOPs) / peak
    wino_fl = sum(agg[k][1] for k in WINO_KINDS if k in agg) / reps
    all_fl = sum(a[1] for a in agg.values()) / reps
    # the 3 x bf16 split kernels' share at their own roof (bf16 peak / 6): the Winograd SPLIT instantiations (their FLOPs / 2.25)
    # and the im2col SPLIT tiles (kinds 651-654: direct multiplies, no / 2.25); everything else at the fp32 MFMA roof
    wsplit_fl = sum(agg[k][1] for k in WINO_SPLIT_KINDS if k in agg) / reps
    gsplit_fl = sum(agg[k][1] for k in IGEMM_SPLIT_KINDS if k in agg) / reps
    floor_ms = ((wino_fl - wsplit_fl) / 2.25 + (all_fl - wino_fl - gsplit_fl)) / (FP32_MFMA_PEAK_TFLOPS * 1e12) * 1e3 + \
               (wsplit_fl / 2.25 + gsplit_fl) / (BF16_MFMA_PEAK_TFLOPS / 6.0 * 1e12) * 1e3
    # algorithmic HBM bytes of the dominant kernel's launches (each input / residual / output tensor and the transformed
    # filters once per launch), from the plan's own launch list
    alg_bytes = None
    try:
        ops = [o for o in plan.op_list(B) if o['kind'] == 50 and o['tile_cfg'] == {455: 11, 465: 11, 555: 12, 565: 12, 575: 13}[dom]
               and (o['h_out'] == 8) == (dom in (465, 565))] if is_wino else []
        if ops:
            tot = 0.0
            for o in ops:
                px = B * o['h_out'] * o['w_out']
                src_px = px // (4 if o['upsample'] else 1)
                tot += 4.0 * (src_px * o['cin'] + px * o['cout'] + 16.0 * o['cin'] * o['cout'])
            alg_bytes = tot / len(ops)
    except Exception:
        pass
    extra = dict(direct_equiv_tflops=achieved, executed_mfma_tflops=executed,
                 step_floor_ms_at_mfma_peaks=floor_ms, launches_ms_per_forward=total_ms,
                 step_frac_of_winograd_roof=(floor_ms / total_ms if total_ms > 0 else None),
                 algorithmic_bytes_per_launch=alg_bytes,
                 traffic_over_algorithmic=(traffic / alg_bytes if traffic and alg_bytes else None),
                 note=('achieved / frac = fp32 multiply-adds the kernel evaluates (direct-conv FLOPs / 2.25 for Winograd F(2x2,3x3)), '
                       'each as six bf16 MFMA products, vs the bf16 MFMA peak / 6 = %.1f TFLOP/s fp32-equivalent; this instantiation is '
                       'bound by its VALU / LDS work (staging, transform, 3-way split), not by the matrix pipe; direct_equiv_tflops = '
                       'SURVEY 8d algorithmic FLOPs / time' % (BF16_MFMA_PEAK_TFLOPS / 6.0)) if (is_split and is_wino) else
                      ('achieved / frac = MFMA FLOPs actually issued (direct-conv FLOPs / 2.25 for Winograd F(2x2,3x3)) vs the fp32 '
                       'MFMA peak; direct_equiv_tflops = SURVEY 8d algorithmic FLOPs / time (can exceed the peak)'))
    return dict(bound='mfma', kernel=names[dom] + (' (Winograd F(2x2,3x3), 6 x v_mfma_f32_32x32x16_bf16 per fp32 product, persistent workgroups)'
                                                   if (is_split and is_wino) else
                                                   (' (6 x v_mfma_f32_32x32x16_bf16 per fp32 product)' if is_split
                                                    else (' (Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32, persistent workgroups)' if is_wino
                                                          else ' (v_mfma_f32_32x32x2_f32)'))), achieved=executed,
                peak=peak, unit='TFLOP/s', frac=executed / peak, traffic=traffic, **extra,
                traffic_note='bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) KB from rocprofv3 --pmc passes, profiles/%s_bench_hbm_pmc.csv'
                             % PROFILE_ROUND,
                avg_launch_us=t_ms / launches * 1e3, rocprof_avg_launch_us=rocprof_avg_us,
                avg_launch_note='avg_launch_us: HIP events around each launch of an eagerly launched forward (includes the launch gap of an '
                                'empty queue, ~5 us); rocprof_avg_launch_us: kernel begin-to-end in profiles/%s_bench_kernel_stats.csv' % PROFILE_ROUND,
                launches_per_forward=launches // reps,
                flops_per_launch=flops / launches, share_of_forward_time=(t_ms / reps) / total_ms,
                all_halo_kernels_tflops=all_tf, sq_counters=counters, by_op_kind=detail)


def split_bf16_leg(netG, st, T, dev, steps=200, option='split_bf16', value=1, restore=None, with_roofline=False):
    """Secondary, NOT the headline: the same reverse step with an opt-in plan option that moves contractions onto
    v_mfma_f32_32x32x16_bf16 with every fp32 operand split into three bf16 terms (six products, fp32 accumulate):
    `split_bf16` (round 1: the direct halo-tile convs with Cout > 64) or `wino_split` (round 4: the Winograd kernel's SPLIT
    instantiation).  Reports its step time and how far its eps is from the exact-fp32 path's on the same input."""
    import torch
    un = netG.denoise_fn
    cond = st['cond']
    shape = tuple(st['img'].shape)
    B = shape[0]
    g = torch.Generator(device=dev).manual_seed(77)
    x = torch.randn(shape, device=dev, generator=g)
    tm = torch.full((B, 1), 0.6, device=dev) if un.variant == 'sr3' else torch.full((B,), 900, dtype=torch.long, device=dev)
    eps_exact = un(x, tm, cond=cond).clone()         # (the plan the headline ran on)
    options = list(option) if isinstance(option, (list, tuple)) else [option]
    # set_option returns the option's previous value: the leg puts back exactly what the headline plan ran with (an A/B run's
    # --plan-opt included), not a constant
    previous = {o: un.plan.set_option(o, value) for o in options}
    roof = None
    try:
        eps_split = un(x, tm, cond=cond).clone()
        st2 = netG._loop_state(shape, None if cond is None else shape, dev)
        if cond is not None:
            st2['cond'].copy_(cond)
        st2['img'].copy_(torch.randn(shape, device=dev))
        st2['step'].fill_(T - 1)
        netG._capture(st2)
        for _ in range(5):
            st2['graph'].replay()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            st2['graph'].replay()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / steps * 1e3
        finite = bool(torch.isfinite(st2['img']).all().item())
        if with_roofline:
            roof = roofline_from_profile(netG, st2['img'], st2['cond'])
    finally:
        for o in options:
            un.plan.set_option(o, previous[o] if restore is None else restore)
        netG._loop_cache = {}
    fl = un.plan.forward_flops(B)
    return dict(ms_per_step=ms, images_per_s_per_gpu=B / (T * ms * 1e-3), step_tflops_equiv=fl / (ms * 1e-3) / 1e12,
                steps=steps, output_finite=finite,
                eps_max_abs_diff_vs_headline_plan=float((eps_split - eps_exact).abs().max().item()),
                eps_max_abs=float(eps_exact.abs().max().item()), roofline=roof,
                note='plan option %s=%d; not used for `value`' % (' / '.join(options), value))


def train_leg(cfg_name, dist, world, rank, dev, batch, steps, warmup):
    """BASELINE.json configs[2] (SR3 16->128, 64 / GPU) or [4] (DDPM-128, 32 / GPU): training step (p_losses + backward +
    Adam, dropout as configured), data parallel with bucketed RCCL all-reduce of the gradients."""
    import numpy as np
    import torch
    if not STUB:
        import model as Model
    c = CONFIGS[cfg_name]
    opt = config_opt(cfg_name, phase='train')
    torch.manual_seed(1000 + rank)              # per-rank streams (z, dropout seed, data); create_model broadcasts rank 0's weights
    np.random.seed(1234 + rank)
    m = _StubModel(cfg_name, dist, dev) if STUB else Model.create_model(opt)
    S = c['size']
    g = torch.Generator().manual_seed(77 + rank)
    data = {'HR': torch.rand(batch, 3, S, S, generator=g) * 2 - 1, 'SR': torch.rand(batch, 3, S, S, generator=g) * 2 - 1}
    m.feed_data(data)
    for _ in range(warmup):
        m.optimize_parameters()
    dev_sync(dev)
    if dist:
        dist.barrier()
    dev_sync(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.optimize_parameters()
    dev_sync(dev)
    if dist:
        dist.barrier()
    dev_sync(dev)
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / steps * 1e3
    fl = 3.0 * m.netG.denoise_fn.plan.forward_flops(batch)
    dp_rec = dp_diagnostics(m, dist, rank, dev) if (dist and not STUB) else None
    return {'roofline': train_roofline(cfg_name, ms), 'data_parallel': dp_rec,
            'metric': '%s training images/sec (p_losses + backward + Adam)' % c['title'], 'value': world * batch / (ms * 1e-3),
            'unit': 'images/s', 'steps_per_s': 1e3 / ms, 'ms_per_step': ms, 'steps': steps, 'warmup': warmup,
            'batch_per_gpu': batch, 'global_batch': batch * world, 'dropout': c['unet']['dropout'],
            'optimizer': 'Adam lr %g' % c['lr'],
            'arithmetic': 'fp32 tensors and accumulation; forward convs, data gradients and (layers with > 64 channels either side) weight '
                          'gradients as six bf16 MFMA products of 3-way split fp32 operands (plan options wino_split / gemm_split / '
                          'wgrad_split, default on; gradients gated against float64 autograd in tests/)',
            'parallelism': 'dp%d, tail-first 32 MB gradient buckets all-reduced (RCCL) as the backward produces them' % world,
            'tflops_at_3x_forward': fl / (ms * 1e-3) / 1e12,
            'frac_of_fp32_mfma_peak': fl / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 'l_pix_last': m.get_current_log()['l_pix'],
            **({'stub': True, 'gradient_buckets_walked': m.buckets_walked,
                'gradient_buckets_per_step': len(m.red.buckets) if m.red else 0} if STUB else {})}


def dp_diagnostics(m, dist, rank, dev, extra_steps=3):
    """After the timed training steps of an N-rank run, per rank: the world size the process group itself reports, the EXPOSED part of
    the gradient all-reduce (event pair on the compute stream around its wait for the reducer's side stream, mean of a few extra
    steps) and a checksum of the parameter arena -- the replicas must be bit-identical (same all-reduced gradient, same Adam step),
    anything else raises.  Gathered to every rank; rank 0 puts the list into the record."""
    import torch
    un = m.netG.denoise_fn
    red = getattr(un, '_reducer', None)
    exposed = None
    if red is not None and red.cuda:
        red.measure_exposed = True
        for _ in range(extra_steps):
            m.optimize_parameters()
        exposed = red.exposed_ms()
        red.measure_exposed = False
    bits = un.arena.data.view(torch.int32).to(torch.int64)
    info = dict(rank=rank, nranks=dist.get_world_size(), backend=dist.get_backend(), device=str(dev),
                exposed_allreduce_ms_per_step=exposed, buckets=None if red is None else len(red.buckets),
                arena_checksum=[int(bits.sum().item()), int((bits * (torch.arange(bits.numel(), device=bits.device) % 8191 + 1)).sum().item())])
    infos = [None] * dist.get_world_size()
    dist.all_gather_object(infos, info)
    same = all(i['arena_checksum'] == infos[0]['arena_checksum'] for i in infos)
    if not same:
        raise RuntimeError('data-parallel replicas diverged after the timed steps: %s' % [i['arena_checksum'] for i in infos])
    return dict(replicas_bit_identical=True, ranks=infos)


def train_roofline(cfg_name, ms_live):
    """`roofline` of the training leg: the training step has no per-launch event table (one engine call), so the dominant kernel, its
    share of the step and its matrix-pipe utilisation come from the committed rocprofv3 passes of the SAME step on the round's final
    tree (profiles/<round>_train_kernel_stats.csv = --kernel-trace --stats of tools/gpu_probe.py --train 64; <round>_train_sq_counters.json
    = its own --pmc SQ pass): frac = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles) = executed MFMA rate / the pipe's roof."""
    if cfg_name != 'sr3_16_128':
        return None
    import csv
    try:
        rows = list(csv.DictReader(open(os.path.join(ROOT, 'profiles', PROFILE_ROUND + '_train_kernel_stats.csv'))))
        rows = [r for r in rows if 'sr3::' in r['Name']]
        tot = sum(float(r['TotalDurationNs']) for r in rows)
        steps = None
        by = sorted(rows, key=lambda r: -float(r['TotalDurationNs']))
        with open(os.path.join(ROOT, 'profiles', PROFILE_ROUND + '_train_sq_counters.json')) as f:
            sq = json.load(f)

        def key(name):
            import re
            return re.sub(r'\(.*$', '', name).replace('void ', '').strip()
        top = []
        for r in by[:6]:
            c = sq.get(key(r['Name']), {})
            top.append({'kernel': key(r['Name']), 'share_of_step_kernel_time': float(r['TotalDurationNs']) / tot,
                        'avg_launch_us': float(r['AverageNs']) / 1e3, 'calls_in_profile': int(r['Calls']),
                        'mfma_busy': c.get('mfma_busy'), 'wait_any': c.get('wait_any'), 'wait_inst': c.get('wait_inst')})
        d = top[0]
        split = ('wgrad_split' in d['kernel']) or ('wino' in d['kernel']) or ('igemm' in d['kernel'])
        peak = BF16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        return {'bound': 'mfma', 'kernel': d['kernel'], 'frac': d['mfma_busy'], 'peak': peak, 'unit': 'TFLOP/s',
                'achieved': (d['mfma_busy'] * peak if d['mfma_busy'] is not None else None),
                'share_of_step_kernel_time': d['share_of_step_kernel_time'], 'avg_launch_us': d['avg_launch_us'],
                'top_kernels': top, 'ms_per_step_live': ms_live,
                'source': 'profiles/%s_train_kernel_stats.csv + %s_train_sq_counters.json (rocprofv3 passes of the same step on the '
                          "round's final tree); achieved = SQ_VALU_MFMA_BUSY_CYCLES fraction x the pipe's dense peak for the kernel's MFMA "
                          'type (bf16 for the 3 x bf16 split kernels: six products per fp32 multiply-add)' % (PROFILE_ROUND, PROFILE_ROUND)}
    except (OSError, KeyError, ValueError, IndexError) as e:
        return {'error': 'no committed training counters for %s: %s' % (PROFILE_ROUND, e)}


def headline_dtype(plan, batch):
    """`dtype` of the line from the plan's EFFECTIVE launch list (an A/B run's --plan-opt included), not from the flags: any
    conv on a 3 x bf16 split instantiation (Winograd 12 / 13, im2col 14-21, the opt-in halo tiles 7 / 8 / 10) -> the split
    label, none -> plain f32."""
    split_tiles = {7, 8, 10, 12, 13} | set(range(14, 22))
    ops = plan.op_list(batch)
    attn = any(o['kind'] == 60 for o in ops) and getattr(plan, 'options', {}).get('attn_split', 1) != 0      # (default on)
    return 'f32 via 3xbf16 split MFMA' if attn or any(o['tile_cfg'] in split_tiles for o in ops) else 'f32'


def build_sampler(cfg_name, B, dev, rank, split_bf16=False, T=2000, exact_fp32=False, plan_opts=None):
    """define_G of a BASELINE.json network (random init, seed 0), its reverse-step hipGraph captured at batch B."""
    import torch
    cfg = CONFIGS[cfg_name]
    if STUB:
        S = cfg['size']
        st = {'img': torch.randn(B, 3, S, S), 'step': torch.zeros(1, dtype=torch.int32), 'cond': None}
        st['graph'] = _StubGraph(st)
        return _StubNet(cfg_name), st
    import model.networks as networks
    torch.manual_seed(0)
    opt = config_opt(cfg_name, T)
    netG = networks.define_G(opt).to(dev)
    netG.set_loss(dev)
    netG.set_new_noise_schedule(opt['model']['beta_schedule']['val'], dev)
    netG.eval()
    netG.denoise_fn.plan.set_option('fuse_stats', 1)
    if split_bf16:
        netG.denoise_fn.plan.set_option('split_bf16', 1)
    if exact_fp32:
        netG.denoise_fn.plan.set_option('wino_split', 0)
        netG.denoise_fn.plan.set_option('gemm_split', 0)
        netG.denoise_fn.plan.set_option('attn_split', 0)
    for k, v in (plan_opts or {}).items():               # A/B runs only (--plan-opt); the record carries them in `config`
        netG.denoise_fn.plan.set_option(k, v)
    S = cfg['size']
    torch.manual_seed(1000 + rank)                       # per-rank RNG stream / inputs
    shape = (B, 3, S, S)
    st = netG._loop_state(shape, shape if cfg['conditional'] else None, dev)
    if cfg['conditional']:
        st['cond'].copy_(torch.rand(shape, device=dev) * 2 - 1)
    st['img'].copy_(torch.randn(shape, device=dev))
    st['step'].fill_(T - 1)
    netG._capture(st)
    return netG, st


STUB = bool(os.environ.get('SR3_BENCH_STUB'))      # CPU / gloo plumbing test (tests/test_dist_gloo.py): the engine is replaced by
                                                   # stand-ins, everything else -- launcher, rank checks, barriers, MAX over ranks,
                                                   # the record, the training leg's bucket walk -- is the code the GPU run takes


def dev_sync(dev):
    import torch
    if dev.type == 'cuda':
        torch.cuda.synchronize(dev)


class _StubGraph(object):
    def __init__(self, st):
        self.st = st

    def replay(self):
        self.st['img'].mul_(0.999)
        self.st['step'].sub_(1)


class _StubPlan(object):
    """Host-only facts of the real plan (parameter table, FLOPs, launch list) -- sr3_plan_create does no device work."""
    def __init__(self, cfg_name):
        from sr3_hip import engine as E
        c = CONFIGS[cfg_name]
        u = c['unet']
        self.p = E.Plan(c['which'], u['in_channel'], u['out_channel'], u['inner_channel'], u.get('norm_groups', 32),
                        u['channel_multiplier'], u['attn_res'], u['res_blocks'], c['size'])
        self.table = self.p.table

    def forward_flops(self, b):
        return self.p.forward_flops(b)

    def op_list(self, b):
        return self.p.op_list(b)


class _StubNet(object):
    def __init__(self, cfg_name):
        self.denoise_fn = type('U', (), {})()
        self.denoise_fn.plan = _StubPlan(cfg_name)
        self._loop_cache = {}


class _StubModel(object):
    """optimize_parameters of the stub: a gradient arena of the real size walked by the real GradReducer (tail-first 32 MB
    buckets, gloo all-reduce), i.e. what model/model.py does behind sr3_train_step."""
    def __init__(self, cfg_name, dist, dev):
        import torch
        from sr3_hip import dist as D
        self.netG = _StubNet(cfg_name)
        n = sum(e['numel'] for e in self.netG.denoise_fn.plan.table)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.red = D.GradReducer(n, dev, dist) if dist else None
        self.buckets_walked = 0
        self.loss = torch.zeros(1)

    def feed_data(self, data):
        self.data = data

    def optimize_parameters(self):
        self.grads.fill_(1.0)
        if self.red:
            self.red.reduce(self.grads, extra=[self.loss])
            self.buckets_walked += len(self.red.buckets)

    def get_current_log(self):
        return {'l_pix': float(self.grads[0])}


def time_replays(st, steps, warmup, T, dist, dev):
    """W untimed + exactly K timed graph replays (chains of T steps), barrier + synchronize on both sides, MAX over ranks."""
    import torch
    graph = st['graph']
    st['step'].fill_(T - 1)
    for _ in range(warmup):
        graph.replay()
    st['step'].fill_(T - 1)
    st['img'].normal_()
    dev_sync(dev)
    if dist:
        dist.barrier()
    dev_sync(dev)
    t0 = time.perf_counter()
    done = 0
    while done < steps:                                   # exactly K steps, chains of T
        n = min(steps - done, T)
        for _ in range(n):
            graph.replay()
        done += n
        if done < steps:
            st['step'].fill_(T - 1)
            st['img'].normal_()
    dev_sync(dev)
    if dist:
        dist.barrier()
    dev_sync(dev)
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


def other_config_leg(cfg_name, dev, steps=50, warmup=3, T=2000, batch=None):
    """A bounded run of another BASELINE.json configuration (configs[3] SR3 64->512 at batch 4, configs[4] DDPM-128 at batch 32)
    for the driver's record: graph-replayed reverse steps, the parity of that graph against the CPU oracle, and the dominant
    kernel's executed-MFMA fraction.  Rank 0, N = 1 only; not part of `value`."""
    import torch
    cfg = CONFIGS[cfg_name]
    B = batch or cfg['batch']
    netG, st = build_sampler(cfg_name, B, dev, 0)
    elapsed = time_replays(st, steps, warmup, T, None, dev)
    ms = elapsed / steps * 1e3
    out = {'metric': '%s images/sec (2000-step sample)' % cfg['title'], 'value': B / (T * ms * 1e-3), 'unit': 'images/s',
           'ms_per_step': ms, 'steps': steps, 'warmup': warmup, 'batch': B,
           'workload': '%s UNet (reference %s; BASELINE.json configs[%d]), batch %d, hipGraph-replayed reverse steps'
                       % (cfg['title'], cfg['ref_json'], cfg['baseline_cfg'], B),
           'output_finite': bool(torch.isfinite(st['img']).all().item())}
    try:
        par = capture_parity_inputs(netG, st, cfg, T)
        rec, _, _ = parity_vs_oracle(cfg_name, netG, par)
        out['parity_max_abs'] = rec['parity_max_abs']
        out['parity_ok'] = rec['ok']
    except Exception as e:
        out['parity_error'] = '%s: %s' % (type(e).__name__, e)
    try:
        rf = roofline_from_profile(netG, st['img'], st['cond'], reps=2)
        out['roofline'] = {k: rf[k] for k in ('kernel', 'achieved', 'peak', 'frac', 'direct_equiv_tflops', 'avg_launch_us',
                                             'launches_per_forward', 'share_of_forward_time', 'step_frac_of_winograd_roof')}
    except Exception as e:
        out['roofline'] = {'error': '%s: %s' % (type(e).__name__, e)}
    st['graph'] = None
    netG._loop_cache = {}
    del netG, st
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------------
def self_launch(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks (one process per GPU, RCCL over 127.0.0.1)."""
    import torch
    have = n if STUB else torch.cuda.device_count()
    if have < n:
        sys.stderr.write('bench.py: --gpus %d but only %d GPU(s) are visible; refusing to report n_gpus=%d\n' % (n, have, n))
        return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, usable_cores() // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='sr3_16_128', choices=sorted(CONFIGS),
                    help='BASELINE.json network: sr3_16_128 (configs[1], the headline; default), sr3_64_512 (configs[3]), ddpm_128 (configs[4])')
    ap.add_argument('--batch', type=int, default=0, help='images per GPU (default: the BASELINE.json batch of --config)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-torch-baseline', action='store_true', help='skip the stock PyTorch-ROCm (MIOpen) leg')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the bounded legs of the other BASELINE.json configurations (SR3 64->512 batch 4, DDPM-128 batch 32)')
    ap.add_argument('--no-split-leg', action='store_true', help='(default now) skip the secondary split_bf16 measurement')
    ap.add_argument('--no-exact-leg', action='store_true',
                    help='skip the secondary measurement with plan options wino_split = gemm_split = 0 (every conv on the fp32 MFMA)')
    ap.add_argument('--exact-fp32', action='store_true',
                    help='run the HEADLINE leg with wino_split = gemm_split = 0 (dtype is then reported as f32)')
    ap.add_argument('--plan-opt', action='append', default=[], metavar='KEY=VALUE',
                    help='A/B runs: set a plan option (sr3_plan_set_option) on the headline leg, e.g. --plan-opt gemm_split=0')
    ap.add_argument('--split-leg', action='store_true',
                    help='also time the opt-in split_bf16 plan option (direct halo kernels on bf16 MFMA; superseded by the fp32 '
                         'Winograd path, which is faster and exact-fp32 arithmetic)')
    ap.add_argument('--split-bf16', action='store_true',
                    help='experiment: run the HEADLINE leg with the split_bf16 plan option (dtype is then reported as '
                         '"f32 via 3xbf16 split MFMA"; the default is the exact-fp32 MFMA path)')
    ap.add_argument('--train-steps', type=int, default=10, help='0 disables the training leg')
    ap.add_argument('--train-batch', type=int, default=0, help='images per GPU (default: the BASELINE.json batch of --config)')
    ap.add_argument('--extra-leg-timeout', type=int, default=600,
                    help='seconds after which the roofline / split / train / baseline legs are abandoned and the line is printed')
    a = ap.parse_args()

    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and a.gpus > 1:
        sys.exit(self_launch(a.gpus))
    world = int(env_world or '1')
    if world != a.gpus and not (a.gpus == 1 and env_world is None):
        sys.stderr.write('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks\n' % (a.gpus, world))
        sys.exit(2)
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))

    sys.path.insert(0, PKG)
    sys.path.insert(0, ROOT)
    import torch
    if not STUB and torch.cuda.device_count() <= local:
        sys.stderr.write('bench.py: rank %d needs cuda:%d but %d device(s) are visible\n' % (rank, local, torch.cuda.device_count()))
        sys.exit(2)
    # Joining the job is the drop-in's own code path (sr3_hip.dist.bootstrap: cuda:LOCAL_RANK, RCCL group bound to it, per-rank
    # RNG) -- the one `model.create_model` / `data.create_dataloader` take under torch.distributed.run
    if os.environ.get('SR3_BENCH_FORCE_DIST'):              # the env knob exercises the collective path on one GPU
        os.environ.setdefault('SR3_DP', 'force')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('RANK', '0')
    if world > 1 or os.environ.get('SR3_BENCH_FORCE_DIST'):
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
    from sr3_hip import dist as D
    b_rank, b_world, b_local = D.bootstrap()
    if world > 1 and (b_rank, b_world, b_local) != (rank, world, local):
        sys.stderr.write('bench.py: bootstrap gave rank %d/%d local %d, the launcher said %d/%d local %d\n'
                         % (b_rank, b_world, b_local, rank, world, local))
        sys.exit(2)
    if STUB:
        dev = torch.device('cpu')
    else:
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    dist = None
    if D.dp_active():
        import torch.distributed as dist

    cfg = CONFIGS[a.config]
    T = 2000
    B = a.batch or cfg['batch']
    S = cfg['size']
    plan_opts = dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in a.plan_opt)
    netG, st = build_sampler(a.config, B, dev, rank, a.split_bf16, T, a.exact_fp32, plan_opts)
    elapsed = time_replays(st, a.steps, a.warmup, T, dist, dev)
    finite = bool(torch.isfinite(st['img']).all().item())
    ms_per_step = elapsed / a.steps * 1e3
    images_per_s = world * B / (T * ms_per_step * 1e-3)

    # ---- the headline record is complete here; everything below is an extra leg that must never cost the line ----
    flops_step = netG.denoise_fn.plan.forward_flops(B)
    nparams = sum(e['numel'] for e in netG.denoise_fn.plan.table)
    rec = {
        'metric': '%s images/sec (2000-step sample)' % cfg['title'], 'value': images_per_s, 'unit': 'images/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        # fp32 tensors and fp32 accumulation everywhere; the Winograd contractions of maps >= 16x16 (the dominant kernel) evaluate
        # each fp32 product as six bf16 MFMA products of 3-way split operands (plan option wino_split, default on, gated in tests/);
        # `exact_fp32` below is the same step with that option off
        'dtype': headline_dtype(netG.denoise_fn.plan, B),
        'data': 'synthetic' if not STUB else 'STUB: CPU plumbing test, no engine -- the numbers mean nothing',
        'config': {'workload': '%s UNet (reference %s; BASELINE.json configs[%d]), batch %d per GPU, 2000-step p_sample_loop via '
                               'hipGraph replay; step = one reverse step of the batch; images/s = n_gpus*batch/(2000*t_step)'
                               % (cfg['title'], cfg['ref_json'], cfg['baseline_cfg'], B),
                   'name': a.config, 'batch_per_gpu': B, 'global_batch': B * world, 'n_timestep': T, 'image_size': S,
                   'params': nparams, 'parallelism': 'independent batches per rank (no collective)',
                   'weights': 'random init (PyTorch default, seed 0)', 'output_finite': finite,
                   **({'plan_options_ab_run': plan_opts} if plan_opts else {})},
        'step_direct_equiv_tflops': flops_step / (ms_per_step * 1e-3) / 1e12,     # SURVEY 8d FLOPs / time (Winograd executes fewer)
        'parity_max_abs': None,
    }
    printed = threading.Lock()

    def emit(note=None):
        if not printed.acquire(False):
            return
        if rank == 0:
            if note:
                rec['note'] = note
            print(json.dumps(rec), flush=True)

    def watchdog():
        # an extra leg hung (e.g. a collective on a sick node): the headline measurement is already done -- print it
        emit('extra legs abandoned after %d s (watchdog); headline fields are complete' % a.extra_leg_timeout)
        os._exit(0)
    timer = threading.Timer(a.extra_leg_timeout, watchdog)
    timer.daemon = True
    timer.start()

    par = None
    if rank == 0 and not STUB:
        try:
            par = capture_parity_inputs(netG, st, cfg, T)
        except Exception as e:
            rec['parity'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if rank == 0 and not a.no_roofline:
        try:
            rec['roofline'] = roofline_from_profile(netG, st['img'], st['cond'])
        except Exception as e:
            rec['roofline'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if rank == 0 and not a.no_exact_leg and not a.split_bf16 and not a.exact_fp32:
        try:       # the same step with every conv on the exact-fp32 MFMA instantiations (plan options wino_split = gemm_split = 0)
            rec['exact_fp32'] = split_bf16_leg(netG, st, T, dev, option=('wino_split', 'gemm_split', 'attn_split'), value=0,
                                               with_roofline=not a.no_roofline)
            rec['exact_fp32']['dtype'] = 'f32 (v_mfma_f32_32x32x2_f32 everywhere)'
        except Exception as e:                     # the secondary leg must never cost the headline line
            rec['exact_fp32'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if rank == 0 and a.split_leg and not a.no_split_leg and not a.split_bf16:
        try:
            rec['split_bf16'] = split_bf16_leg(netG, st, T, dev)
        except Exception as e:
            rec['split_bf16'] = {'error': '%s: %s' % (type(e).__name__, e)}
    # free the sampling state (graph, workspace) -- the training workspace is ~18 GB at batch 64
    st['graph'] = None
    netG._loop_cache = {}
    torch.cuda.empty_cache()
    if a.train_steps > 0:
        try:
            train = train_leg(a.config, dist, world, rank, dev, a.train_batch or cfg['train_batch'], a.train_steps, 2)
        except Exception as e:                      # the headline line must survive a failing extra leg
            train = {'error': '%s: %s' % (type(e).__name__, e)}
        rec['train'] = train
        torch.cuda.empty_cache()
    if rank == 0 and par is not None and not a.no_cpu_baseline:
        try:
            if world == 1:
                cb = cpu_baseline(a.config, netG, par)
                rec['parity'] = cb.pop('_parity')
                rec['cpu_baseline'] = cb
            else:                                   # N > 1: the parity check only (baselines are an N = 1 leg)
                rec['parity'] = parity_vs_oracle(a.config, netG, par)[0]
            rec['parity_max_abs'] = rec['parity']['parity_max_abs']
        except Exception as e:
            rec['cpu_baseline'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if rank == 0 and world == 1 and par is not None and not a.no_torch_baseline:
        try:
            rec['torch_rocm_baseline'] = torch_rocm_baseline(a.config, netG, par, dev)
        except Exception as e:
            rec['torch_rocm_baseline'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if rank == 0 and world == 1 and a.config == 'sr3_16_128' and not a.no_other_configs:
        # the other BASELINE.json configurations, bounded (50 replayed steps each), so that they are in the driver's record too
        rec['other_configs'] = {}
        for name in ('sr3_64_512', 'ddpm_128'):
            try:
                rec['other_configs'][name] = other_config_leg(name, dev)
            except Exception as e:
                rec['other_configs'][name] = {'error': '%s: %s' % (type(e).__name__, e)}
            torch.cuda.empty_cache()
        # the batch the reference's own infer.py runs (validation loader batch_size = 1, data/__init__.py:18; one 2000-step chain per
        # image, infer.py:67-71): the launch-bound end of the same graph
        try:
            rec['other_configs']['sr3_16_128_b1'] = other_config_leg('sr3_16_128', dev, batch=1)
            rec['other_configs']['sr3_16_128_b1']['note'] = ('batch 1 = what the reference infer.py feeds per call; sr3_hip.dist.ValWave batches '
                                                             'consecutive validation items into one chain batch instead (INTEGRATION.md)')
        except Exception as e:
            rec['other_configs']['sr3_16_128_b1'] = {'error': '%s: %s' % (type(e).__name__, e)}
        torch.cuda.empty_cache()
        # BASELINE.json configs[4] as a TRAINING workload (DDPM-128, batch 32 / GPU, dropout 0.2): a bounded leg of 5 steps
        try:
            rec['other_configs']['ddpm_128_train'] = train_leg('ddpm_128', None, 1, 0, dev, CONFIGS['ddpm_128']['train_batch'], 5, 2)
        except Exception as e:
            rec['other_configs']['ddpm_128_train'] = {'error': '%s: %s' % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    timer.cancel()
    emit()
    if dist:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == '__main__':
    main()
