#!/usr/bin/env python3
"""bench.py -- SR3 16->128 sampling throughput of the MI355X engine (BASELINE.json metric).

Workload (BASELINE.json configs[1]): the SR3 16->128 UNet of config/sr_sr3_16_128.json (inner 64,
mults 1,2,4,8,8, attention at 16x16, 97.8 M fp32 parameters, random init), batch 16 per GPU,
T = 2000 linear-beta reverse steps.  A *step* is one reverse step p_sample of the whole batch:
[z ~ N(0,I)] -> UNet forward -> fused x_{t-1} update -> counter decrement, replayed from one
hipGraph.  The default --steps 2000 times one complete sample; images/s = N * B / (T * t_step).

One JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (the halo-tile 3x3 conv on v_mfma_f32_32x32x2_f32): algorithmic
                  FLOPs per launch / average launch duration, measured with HIP events around every
                  launch of the plan (sr3_unet_forward_profile) right after the timed region.
  cpu_baseline -- the CPU oracle (oracle/sr3_oracle.py, a port of the reference's algorithm) timed on
                  this node's host cores on a bounded sample (a few reverse steps at the same batch).
Multi-GPU: one process per GPU (torch.distributed.run), independent image batches per rank (the
reverse chains share nothing), no collective in the data path; the timed region is bracketed by
barrier + synchronize and the MAX over ranks is reported ("scaling": "weak").
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd'))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (~2.5 PF)


def sr3_16_128_opt(n_timestep=2000):
    """The `model` subtree of the reference's config/sr_sr3_16_128.json (lines 39-77)."""
    sched = dict(schedule='linear', n_timestep=n_timestep, linear_start=1e-6, linear_end=1e-2)
    return {
        'phase': 'val', 'gpu_ids': [0], 'distributed': False,
        'path': {'checkpoint': '/tmp', 'resume_state': None},
        'train': {'optimizer': {'type': 'adam', 'lr': 1e-4}},
        'model': {
            'which_model_G': 'sr3', 'finetune_norm': False,
            'unet': dict(in_channel=6, out_channel=3, inner_channel=64, channel_multiplier=[1, 2, 4, 8, 8],
                         attn_res=[16], res_blocks=2, dropout=0.2),
            'beta_schedule': {'train': dict(sched), 'val': dict(sched)},
            'diffusion': dict(image_size=128, channels=3, conditional=True),
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


def cpu_baseline(batch, budget_s=25.0):
    """Oracle p_sample (UNet forward + update) on the host cores, bounded sample."""
    from oracle import sr3_oracle as O
    torch.manual_seed(0)
    opt = sr3_16_128_opt()
    desc = O.desc_from_opt(opt)
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    # random-init weights with the reference's shapes (values do not matter for timing)
    from sr3_hip import engine as E
    plan = E.Plan('sr3', 6, 3, 64, 32, [1, 2, 4, 8, 8], [16], 2, 128)
    sd = {}
    for e in plan.table:
        sd['denoise_fn.' + e['name']] = torch.randn(e['shape']) * 0.02
    tab = O.schedule_tables(opt['model']['beta_schedule']['val'])
    x = torch.randn(batch, 3, 128, 128)
    sr = torch.rand(batch, 3, 128, 128) * 2 - 1
    z = torch.randn(batch, 3, 128, 128)
    times = []
    with torch.no_grad():
        t0 = time.time()
        O.p_sample(sd, desc, tab, x[:1], 1999, z[:1], condition_x=sr[:1])   # warm-up (1 image: allocator, oneDNN)
        warm = time.time() - t0
        n = 0
        start = time.time()
        while n < 8 and (n == 0 or (time.time() - start) * (n + 1) / n + warm < budget_s):
            t1 = time.time()
            x = O.p_sample(sd, desc, tab, x, 1998 - n, z, condition_x=sr)
            times.append(time.time() - t1)
            n += 1
    if not times:
        times = [warm]
    t_step = sum(times) / len(times)
    return dict(value=batch / (2000.0 * t_step), unit='images/s', cores=int(torch.get_num_threads()), kind='port',
                sample='%d reverse steps (oracle p_sample: UNet forward + update) at batch %d after 1 warm-up, '
                       '%.2f s/step, extrapolated x2000' % (len(times), batch, t_step))


def roofline_from_profile(netG, x, cond, level, reps=3):
    """HIP-event timing of every launch of one forward; aggregates the dominant kernel."""
    from sr3_hip import lib as L
    un = netG.denoise_fn
    plan = un.plan
    lib = L.load()
    B = x.shape[0]
    wsbuf, need = un._ws.get(plan, B, x.device)
    out = torch.empty(B, 3, 128, 128, device=x.device)
    max_ops = 4096
    ms = (C.c_float * max_ops)()
    kind = (C.c_int * max_ops)()
    fl = (C.c_double * max_ops)()
    n = C.c_int()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    agg = {}
    for r in range(reps + 1):
        L.check(lib.sr3_unet_forward_profile(plan.handle, L.ptr(x), L.ptr(cond), 3, L.ptr(level), None, L.ptr(un.freq),
                                             L.ptr(un.arena.data), L.ptr(wsbuf), need, L.ptr(out), B, stream, max_ops,
                                             ms, kind, fl, C.byref(n)))
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
             355: 'k_conv3x3_halo<4,2,false,false,1,2>', 357: 'k_conv3x3_halo<4,2,true,false,1,2>'}
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
    try:        # HBM bytes per launch from the committed rocprofv3 PMC passes of this command (profiles/)
        with open(os.path.join(ROOT, 'profiles', 'r01_hbm_traffic.json')) as f:
            traffic = json.load(f)['sr3::' + names[dom].replace(',', ', ')]['hbm_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        pass
    is_split = names[dom].split(',')[4] == '1'
    # split kernels: six bf16 MFMA products per fp32 product -> fp32-equivalent peak = bf16 dense peak / 6
    peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if is_split else FP32_MFMA_PEAK_TFLOPS
    return dict(bound='mfma', kernel=names[dom] + (' (6 x v_mfma_f32_32x32x16_bf16 per fp32 product)' if is_split
                                                   else ' (v_mfma_f32_32x32x2_f32)'), achieved=achieved,
                peak=peak, unit='TFLOP/s', frac=achieved / peak, traffic=traffic,
                traffic_note='bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) KB from rocprofv3 --pmc passes, profiles/r01_bench_hbm_pmc.csv',
                avg_launch_us=t_ms / launches * 1e3, launches_per_forward=launches // reps,
                flops_per_launch=flops / launches, share_of_forward_time=(t_ms / reps) / total_ms,
                all_halo_kernels_tflops=all_tf, by_op_kind=detail)


def split_bf16_leg(netG, cond, T, dev, steps=200):
    """Secondary, NOT the headline: the same reverse step with the opt-in `split_bf16` plan option (halo-tile
    convs with Cout > 64 on v_mfma_f32_32x32x16_bf16, each fp32 operand split into three bf16 terms, six products,
    fp32 accumulate).  Reports its step time and how far its eps is from the exact-fp32 path's on the same input."""
    un = netG.denoise_fn
    B = cond.shape[0]
    g = torch.Generator(device=dev).manual_seed(77)
    x = torch.randn(B, 3, 128, 128, device=dev, generator=g)
    level = torch.full((B, 1), 0.6, device=dev)
    eps_exact = un(x, level, cond=cond).clone()
    un.plan.set_option('split_bf16', 1)
    try:
        eps_split = un(x, level, cond=cond).clone()
        shape = tuple(cond.shape)
        st = netG._loop_state(shape, shape, dev)
        st['cond'].copy_(cond)
        st['img'].copy_(torch.randn(shape, device=dev))
        st['step'].fill_(T - 1)
        netG._capture(st)
        for _ in range(5):
            st['graph'].replay()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            st['graph'].replay()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / steps * 1e3
        finite = bool(torch.isfinite(st['img']).all().item())
    finally:
        un.plan.set_option('split_bf16', 0)
        netG._loop_cache = {}
    fl = un.plan.forward_flops(B)
    return dict(ms_per_step=ms, images_per_s_per_gpu=B / (T * ms * 1e-3), step_tflops_equiv=fl / (ms * 1e-3) / 1e12,
                steps=steps, output_finite=finite,
                eps_max_abs_diff_vs_exact_fp32=float((eps_split - eps_exact).abs().max().item()),
                eps_max_abs=float(eps_exact.abs().max().item()),
                note='opt-in plan option split_bf16=1; not used for `value`')


def train_leg(dist, world, rank, dev, batch, steps, warmup):
    """BASELINE.json configs[2]: SR3 16->128 training step (p_losses + backward + Adam, dropout 0.2 as
    configured), `batch` images per GPU, data parallel with bucketed RCCL all-reduce of the gradients."""
    import numpy as np
    import model as Model
    opt = sr3_16_128_opt()
    opt['phase'] = 'train'
    torch.manual_seed(0)
    np.random.seed(1234 + rank)
    m = Model.create_model(opt)
    g = torch.Generator().manual_seed(77 + rank)
    data = {'HR': torch.rand(batch, 3, 128, 128, generator=g) * 2 - 1, 'SR': torch.rand(batch, 3, 128, 128, generator=g) * 2 - 1}
    m.feed_data(data)
    for _ in range(warmup):
        m.optimize_parameters()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.optimize_parameters()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / steps * 1e3
    fl = 3.0 * m.netG.denoise_fn.plan.forward_flops(batch)
    return {'metric': 'SR3 16->128 training images/sec (p_losses + backward + Adam)', 'value': world * batch / (ms * 1e-3),
            'unit': 'images/s', 'steps_per_s': 1e3 / ms, 'ms_per_step': ms, 'steps': steps, 'warmup': warmup,
            'batch_per_gpu': batch, 'global_batch': batch * world, 'dropout': 0.2, 'optimizer': 'Adam lr 1e-4',
            'parallelism': 'dp%d, tail-first 32 MB gradient buckets all-reduced (RCCL) as the backward produces them' % world,
            'tflops_at_3x_forward': fl / (ms * 1e-3) / 1e12,
            'frac_of_fp32_mfma_peak': fl / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 'l_pix_last': m.get_current_log()['l_pix']}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16, help='images per GPU (BASELINE config: 16)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-split-leg', action='store_true', help='skip the secondary split_bf16 measurement')
    ap.add_argument('--split-bf16', action='store_true',
                    help='experiment: run the HEADLINE leg with the split_bf16 plan option (dtype is then reported as '
                         '"f32 via 3xbf16 split MFMA"; the default is the exact-fp32 MFMA path)')
    ap.add_argument('--train-steps', type=int, default=10, help='0 disables the training leg')
    ap.add_argument('--train-batch', type=int, default=64, help='images per GPU (BASELINE config: 64)')
    ap.add_argument('--extra-leg-timeout', type=int, default=420,
                    help='seconds after which the roofline / split / train / cpu legs are abandoned and the line is printed')
    a = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)              # before the process group: RCCL binds its communicator to the current device
    dev = torch.device('cuda', local)
    dist = None
    if world > 1 or os.environ.get('SR3_BENCH_FORCE_DIST'):     # the env knob exercises the collective path on one GPU
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    import model.networks as networks
    T = 2000
    torch.manual_seed(0)
    opt = sr3_16_128_opt(T)
    netG = networks.define_G(opt).to(dev)
    netG.set_loss(dev)
    netG.set_new_noise_schedule(opt['model']['beta_schedule']['val'], dev)
    netG.eval()
    netG.denoise_fn.plan.set_option('fuse_stats', 1)
    if a.split_bf16:
        netG.denoise_fn.plan.set_option('split_bf16', 1)
    B = a.batch
    torch.manual_seed(1000 + rank)                       # per-rank RNG stream / inputs
    cond = (torch.rand(B, 3, 128, 128, device=dev) * 2 - 1)
    shape = (B, 3, 128, 128)
    st = netG._loop_state(shape, shape, dev)
    st['cond'].copy_(cond)
    st['img'].copy_(torch.randn(shape, device=dev))
    st['step'].fill_(T - 1)
    netG._capture(st)
    graph = st['graph']

    st['step'].fill_(T - 1)
    # warm-up
    for _ in range(a.warmup):
        graph.replay()
    st['step'].fill_(T - 1)
    st['img'].normal_()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    done = 0
    while done < a.steps:                                 # exactly K steps, chains of T
        n = min(a.steps - done, T)
        for _ in range(n):
            graph.replay()
        done += n
        if done < a.steps:
            st['step'].fill_(T - 1)
            st['img'].normal_()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = bool(torch.isfinite(st['img']).all().item())
    ms_per_step = elapsed / a.steps * 1e3
    images_per_s = world * B / (T * ms_per_step * 1e-3)

    # ---- the headline record is complete here; everything below is an extra leg that must never cost the line ----
    flops_step = netG.denoise_fn.plan.forward_flops(B)
    rec = {
        'metric': 'SR3 16->128 images/sec (2000-step sample)', 'value': images_per_s, 'unit': 'images/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 via 3xbf16 split MFMA' if a.split_bf16 else 'f32', 'data': 'synthetic',
        'config': {'workload': 'SR3 16->128 UNet (reference config/sr_sr3_16_128.json), batch %d per GPU, '
                               '2000-step p_sample_loop via hipGraph replay; step = one reverse step of the batch; '
                               'images/s = n_gpus*batch/(2000*t_step)' % B,
                   'batch_per_gpu': B, 'global_batch': B * world, 'n_timestep': T, 'image_size': 128,
                   'params': 97807491, 'parallelism': 'independent batches per rank (no collective)',
                   'weights': 'random init (PyTorch default, seed 0)', 'output_finite': finite},
        'step_tflops': flops_step / (ms_per_step * 1e-3) / 1e12,
        'step_frac_of_fp32_mfma_peak': flops_step / (ms_per_step * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
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

    if rank == 0 and not a.no_roofline:
        level = torch.full((B,), 0.5, device=dev)
        try:
            rec['roofline'] = roofline_from_profile(netG, st['img'], st['cond'], level)
        except Exception as e:
            rec['roofline'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if rank == 0 and not a.no_split_leg and not a.split_bf16:
        try:
            rec['split_bf16'] = split_bf16_leg(netG, st['cond'], T, dev)
        except Exception as e:
            rec['split_bf16'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if a.train_steps > 0:
        # free the sampling state first (graph, workspace) -- the training workspace is ~18 GB at batch 64
        st['graph'] = None
        netG._loop_cache = {}
        torch.cuda.empty_cache()
        try:
            train = train_leg(dist, world, rank, dev, a.train_batch, a.train_steps, 2)
        except Exception as e:                      # the headline line must survive a failing extra leg
            train = {'error': '%s: %s' % (type(e).__name__, e)}
        rec['train'] = train
    if rank == 0 and not a.no_cpu_baseline and world == 1:
        try:
            rec['cpu_baseline'] = cpu_baseline(B)
        except Exception as e:
            rec['cpu_baseline'] = {'error': '%s: %s' % (type(e).__name__, e)}
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
