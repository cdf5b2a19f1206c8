#!/usr/bin/env python3
"""GPU probe: per-layer conv sweep (tile config x split-K) on the real SR3 16->128 shapes and a
full UNet forward timing.  Writes JSON lines to gpurun_out/probe_*.jsonl.  Not part of the product."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd'))
from sr3_hip import lib as L, engine as E      # noqa: E402

OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)


def time_fn(fn, warm=2, iters=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def conv_sweep(B, out_path, quick=False, only=None, cfgs=(1, 2, 3, 4), kss=(1, 2, 4, 8)):
    lib = L.load()
    d = torch.device('cuda:0')
    # (name, C0, C1, H (source), Cout, k, stride, ups, act)
    shapes = [
        ('128_64_64', 64, 0, 128, 64, 3, 1, 0, 2),
        ('128_128_64', 64, 64, 128, 64, 3, 1, 0, 2),
        ('128_192_64', 128, 64, 128, 64, 3, 1, 0, 2),
        ('up64_128', 128, 0, 64, 128, 3, 1, 1, 0),
        ('down128_64', 64, 0, 128, 64, 3, 2, 0, 0),
        ('64_128_128', 128, 0, 64, 128, 3, 1, 0, 2),
        ('64_384_128', 256, 128, 64, 128, 3, 1, 0, 2),
        ('32_256_256', 256, 0, 32, 256, 3, 1, 0, 2),
        ('32_768_256', 512, 256, 32, 256, 3, 1, 0, 2),
        ('16_512_512', 512, 0, 16, 512, 3, 1, 0, 2),
        ('16_1024_512', 512, 512, 16, 512, 3, 1, 0, 2),
        ('8_512_512', 512, 0, 8, 512, 3, 1, 0, 2),
        ('8_1024_512', 512, 512, 8, 512, 3, 1, 0, 2),
        ('k1_16_512_1536', 512, 0, 16, 1536, 1, 1, 0, 1),
        ('k1_16_512_512', 512, 0, 16, 512, 1, 1, 0, 0),
        ('k1_16_1024_512', 512, 512, 16, 512, 1, 1, 0, 0),
        ('k1_128_192_64', 128, 64, 128, 64, 1, 1, 0, 0),
    ]
    if only and 'net1x1' in only:
        # every 1x1 / stride-2 conv shape of the C2 forward (tools/op_table.py): (name, C0, C1, H, Cout, k, stride, ups, act)
        shapes = [('k1_64_64_128', 64, 0, 64, 128, 1, 1, 0, 0), ('k1_32_128_256', 128, 0, 32, 256, 1, 1, 0, 0),
                  ('k1_16_256_512', 256, 0, 16, 512, 1, 1, 0, 0), ('k1_16_512_1536', 512, 0, 16, 1536, 1, 1, 0, 1),
                  ('k1_16_512_512', 512, 0, 16, 512, 1, 1, 0, 0), ('k1_8_512_1536', 512, 0, 8, 1536, 1, 1, 0, 1),
                  ('k1_8_512_512', 512, 0, 8, 512, 1, 1, 0, 0), ('k1_8_1024_512', 512, 512, 8, 512, 1, 1, 0, 0),
                  ('k1_16_1024_512', 512, 512, 16, 512, 1, 1, 0, 0), ('k1_16_768_512', 512, 256, 16, 512, 1, 1, 0, 0),
                  ('k1_32_768_256', 512, 256, 32, 256, 1, 1, 0, 0), ('k1_32_512_256', 256, 256, 32, 256, 1, 1, 0, 0),
                  ('k1_32_384_256', 256, 128, 32, 256, 1, 1, 0, 0), ('k1_64_384_128', 256, 128, 64, 128, 1, 1, 0, 0),
                  ('k1_64_256_128', 128, 128, 64, 128, 1, 1, 0, 0), ('k1_64_192_128', 128, 64, 64, 128, 1, 1, 0, 0),
                  ('k1_128_192_64', 128, 64, 128, 64, 1, 1, 0, 0), ('k1_128_128_64', 64, 64, 128, 64, 1, 1, 0, 0),
                  ('s2_128_64', 64, 0, 128, 64, 3, 2, 0, 0), ('s2_64_128', 128, 0, 64, 128, 3, 2, 0, 0),
                  ('s2_32_256', 256, 0, 32, 256, 3, 2, 0, 0), ('s2_16_512', 512, 0, 16, 512, 3, 2, 0, 0)]
    elif only:
        shapes = [s_ for s_ in shapes if s_[0] in only]
    if quick:
        shapes = shapes[:2] + shapes[7:8] + shapes[9:10] + shapes[11:12]
    with open(out_path, 'w') as f:
        for (name, C0, C1, H, Cout, k, stride, ups, act) in shapes:
            Cin = C0 + C1
            pad = k // 2
            Ho = ((H << ups) + 2 * pad - k) // stride + 1
            s0 = torch.randn(B, H, H, C0, device=d)
            s1 = torch.randn(B, H, H, C1, device=d) if C1 else None
            w = torch.randn(Cout, k * k, Cin, device=d) * 0.02
            bias = torch.randn(Cout, device=d)
            ss = torch.randn(B, Cin, 2, device=d) if act else None
            out = torch.empty(B, Ho, Ho, Cout, device=d)
            flops = 2.0 * B * Ho * Ho * Cout * Cin * k * k
            total_it = ((Cin + 31) // 32) * k * k
            for cfg in cfgs:
                if cfg in (1, 4, 14, 17) and Cout <= 64:
                    continue
                if 5 <= cfg <= 13 and (k != 3 or stride != 1):
                    continue
                for ks in kss:
                    if ks > 1 and (ks * 4 > total_it):
                        continue
                    if 5 <= cfg <= 13 and ks > 1 and ks * 2 > (Cin + 31) // 32:
                        continue
                    bm, bn = {0: (128, 128), 1: (128, 128), 2: (128, 64), 3: (64, 64), 4: (64, 128), 5: (128, 128), 6: (256, 64), 7: (128, 128), 8: (256, 64), 9: (256, 128), 10: (256, 128), 11: (256, 64), 12: (128, 128), 14: (128, 128), 15: (128, 64), 16: (64, 64), 17: (64, 128)}[cfg]
                    tiles = -(-B * Ho * Ho // bm) * -(-Cout // bn)
                    if ks > 1 and tiles * ks > 4096:
                        continue
                    nb = int(lib.sr3_conv_scratch_bytes(B, Ho, Ho, Cin, Cout, k, cfg, ks))
                    scratch = torch.empty(max(nb, 16), dtype=torch.uint8, device=d)
                    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

                    def run():
                        L.check(lib.sr3_conv_f32(L.ptr(s0), C0, L.ptr(s1), C1, B, H, H, ups, stride, k, Cout, L.ptr(w),
                                                 L.ptr(bias), L.ptr(ss), act, None, 0, None, 0, None, 0, L.ptr(out),
                                                 None, cfg, ks, L.ptr(scratch), nb, st))
                    print('run', name, cfg, ks, nb, flush=True)
                    try:
                        ms = time_fn(run)
                    except L.Sr3Error as e:
                        print('skip', name, cfg, ks, str(e)[:80], flush=True)
                        continue
                    rec = dict(shape=name, B=B, cfg=cfg, ksplit=ks, tiles=tiles, ms=ms, tflops=flops / ms / 1e9)
                    f.write(json.dumps(rec) + '\n')
                    f.flush()
                    print(rec, flush=True)
            del s0, s1, w, out, ss


def unet_time(B, out_path, fuse, split_bf16=0):
    d = torch.device('cuda:0')
    plan = E.Plan('sr3', 6, 3, 64, 32, [1, 2, 4, 8, 8], [16], 2, 128)
    plan.set_option('fuse_stats', fuse)
    plan.set_option('split_bf16', split_bf16)
    torch.manual_seed(0)
    arena = torch.randn(plan.param_floats, device=d) * 0.02
    freq = plan.default_freq().to(d)
    ws = E.Workspace()
    x = torch.randn(B, 3, 128, 128, device=d)
    cond = torch.randn(B, 3, 128, 128, device=d)
    lvl = torch.full((B,), 0.5, device=d)
    out = torch.empty(B, 3, 128, 128, device=d)
    fn = lambda: E.unet_forward(plan, arena, freq, ws, x, cond=cond, noise_level=lvl, out=out)
    ms = time_fn(fn, warm=2, iters=5)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    msg = time_fn(g.replay, warm=2, iters=5)
    fl = plan.forward_flops(B)
    ref = getattr(unet_time, '_ref', None)
    err = None
    if ref is not None and ref.shape == out.shape:
        err = float((out - ref).abs().max() / ref.abs().max())
    else:
        unet_time._ref = out.clone()
    rec = dict(what='unet_forward', B=B, fuse_stats=fuse, split_bf16=split_bf16, rel_err_vs_first=err, ms_eager=ms, ms_graph=msg, tflops_graph=fl / msg / 1e9,
               ops=plan.num_ops(B), ws_gb=plan.workspace_bytes(B) / 1e9, finite=bool(torch.isfinite(out).all()))
    with open(out_path, 'a') as f:
        f.write(json.dumps(rec) + '\n')
    print(rec, flush=True)


def op_compare(B, out_path, reps=3):
    """Per-op HIP-event times of one forward: exact-fp32 plan vs split_bf16 plan, side by side."""
    import ctypes as C
    from sr3_hip import lib as L
    lib = L.load()
    d = torch.device('cuda:0')
    res = {}
    for mode in (0, 1):
        plan = E.Plan('sr3', 6, 3, 64, 32, [1, 2, 4, 8, 8], [16], 2, 128)
        plan.set_option('split_bf16', mode)
        torch.manual_seed(0)
        arena = torch.randn(plan.param_floats, device=d) * 0.02
        freq = plan.default_freq().to(d)
        x = torch.randn(B, 3, 128, 128, device=d)
        cond = torch.randn(B, 3, 128, 128, device=d)
        lvl = torch.full((B,), 0.5, device=d)
        out = torch.empty(B, 3, 128, 128, device=d)
        need = plan.workspace_bytes(B)
        ws = torch.empty(need + 256, dtype=torch.uint8, device=d)
        wsp = ws.data_ptr() + (-ws.data_ptr()) % 256
        n = C.c_int()
        ms = (C.c_float * 4096)(); kind = (C.c_int * 4096)(); fl = (C.c_double * 4096)()
        acc = None
        for r in range(reps + 1):
            L.check(lib.sr3_unet_forward_profile(plan.handle, L.ptr(x), L.ptr(cond), 3, L.ptr(lvl), None, L.ptr(freq),
                                                 L.ptr(arena), C.c_void_p(wsp), need, L.ptr(out), B,
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream), 4096, ms, kind, fl,
                                                 C.byref(n)))
            if r == 0:
                acc = [0.0] * n.value
                continue
            for i in range(n.value):
                acc[i] += ms[i] / reps
        res[mode] = [(kind[i], fl[i], acc[i]) for i in range(n.value)]
    with open(out_path, 'w') as f:
        tot = [0.0, 0.0]
        for (k0, f0, m0), (k1, f1, m1) in zip(res[0], res[1]):
            tot[0] += m0; tot[1] += m1
            if k0 != k1:
                f.write('%4d %4d gflop %8.2f  fp32 %7.1f us %6.1f TF | split %7.1f us %6.1f TF  x%.2f\n' % (
                    k0, k1, f0 / 1e9, m0 * 1e3, f0 / max(m0, 1e-9) / 1e9, m1 * 1e3, f1 / max(m1, 1e-9) / 1e9, m0 / max(m1, 1e-9)))
        f.write('total fp32 %.3f ms split %.3f ms\n' % (tot[0], tot[1]))
    print(open(out_path).read(), flush=True)


def io_times(out_path):
    """HBM-bound helpers either side of the path: time and effective bandwidth."""
    import core.metrics as M
    import data.util as U
    d = torch.device('cuda:0')
    recs = []
    for (B, S) in ((64, 128), (256, 128), (16, 512), (64, 512)):
        u8 = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device=d)
        flip = torch.randint(0, 2, (B,), dtype=torch.uint8, device=d)
        out = torch.empty(B, 3, S, S, device=d)
        ms = time_fn(lambda: U.u8_batch_to_f32(u8, flip, (-1, 1), d, out=out), warm=3, iters=20)
        by = B * S * S * 3 * 5
        recs.append(dict(what='images_u8_to_f32', B=B, S=S, us=ms * 1e3, algorithmic_bytes=by, GBps=by / ms / 1e6))
        x = torch.randn(B, 3, S, S, device=d)
        y = (x + 0.05 * torch.randn_like(x))
        lib = L.load()
        nb = int(lib.sr3_eval_scratch_bytes(B, 3, S, S))
        scratch = torch.empty(nb + 256, dtype=torch.uint8, device=d)
        sse = torch.empty(B, dtype=torch.int64, device=d)
        ssim = torch.empty(B, dtype=torch.float64, device=d)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        off = (-scratch.data_ptr()) % 256

        def ev():
            L.check(lib.sr3_eval_psnr_ssim_f32(L.ptr(x), L.ptr(y), B, 3, S, S, -1.0, 1.0, C.c_void_p(scratch.data_ptr() + off), nb,
                                               L.ptr(sse), L.ptr(ssim), st))
        ms = time_fn(ev, warm=3, iters=20)
        by = B * S * S * 3 * 8          # two fp32 images read once
        fl = B * (S - 10) * (S - 10) * 3 * 121 * 10.0
        recs.append(dict(what='eval_psnr_ssim_f32', B=B, S=S, us=ms * 1e3, algorithmic_bytes=by, GBps=by / ms / 1e6,
                         f64_gflops=fl / ms / 1e6))
        t = time_fn(lambda: M.tensor2img_device(x), warm=3, iters=20)
        by = B * S * S * 3 * 5
        recs.append(dict(what='tensor2img(grid)', B=B, S=S, us=t * 1e3, algorithmic_bytes=by, GBps=by / t / 1e6))
    with open(out_path, 'w') as f:
        for r in recs:
            f.write(json.dumps(r) + '\n')
            print(r, flush=True)


def unet_time_split(B, nsplit, out_path):
    """One forward of B images as `nsplit` independent graph branches of B/nsplit images each."""
    d = torch.device('cuda:0')
    plan = E.Plan('sr3', 6, 3, 64, 32, [1, 2, 4, 8, 8], [16], 2, 128)
    arena = torch.randn(plan.param_floats, device=d) * 0.02
    freq = plan.default_freq().to(d)
    Bs = B // nsplit
    wss = [E.Workspace() for _ in range(nsplit)]
    x = torch.randn(B, 3, 128, 128, device=d)
    cond = torch.randn(B, 3, 128, 128, device=d)
    lvl = torch.full((B,), 0.5, device=d)
    out = torch.empty(B, 3, 128, 128, device=d)
    streams = [torch.cuda.Stream(d) for _ in range(nsplit)]

    def fn():
        cur = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                sl = slice(i * Bs, (i + 1) * Bs)
                E.unet_forward(plan, arena, freq, wss[i], x[sl], cond=cond[sl], noise_level=lvl[sl], out=out[sl])
        for s in streams:
            cur.wait_stream(s)
    ms = time_fn(fn, warm=2, iters=5)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    msg = time_fn(g.replay, warm=2, iters=10)
    fl = plan.forward_flops(Bs) * nsplit
    rec = dict(what='unet_forward_split', B=B, nsplit=nsplit, ms_eager=ms, ms_graph=msg, tflops_graph=fl / msg / 1e9,
               finite=bool(torch.isfinite(out).all()))
    with open(out_path, 'a') as f:
        f.write(json.dumps(rec) + '\n')
    print(rec, flush=True)


def config_times(out_path):
    """Forward timing (hipGraph) of the other BASELINE.json configurations."""
    d = torch.device('cuda:0')
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import model.networks as networks
    for (name, cfg, B) in [('C2 sr3_16_128 B16', 'sr3_16_128', 16), ('C4 sr3_64_512 B4', 'sr3_64_512', 4),
                           ('C5 ddpm_128 B32', 'ddpm_128', 32), ('C2 sr3_16_128 B1', 'sr3_16_128', 1)]:
        c = bench.CONFIGS[cfg]
        torch.manual_seed(0)
        netG = networks.define_G(bench.config_opt(cfg)).to(d)
        un = netG.denoise_fn
        plan = un.plan
        size, cc = c['size'], 3 if c['conditional'] else 0
        x = torch.randn(B, 3, size, size, device=d)
        cond = torch.randn(B, cc, size, size, device=d) if cc else None
        out = torch.empty(B, 3, size, size, device=d)
        tm = torch.full((B,), 0.5, device=d) if c['which'] == 'sr3' else torch.full((B,), 777, device=d, dtype=torch.long)
        fn = lambda: un(x, tm, cond=cond, out=out)
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        ms = time_fn(g.replay, warm=2, iters=5)
        fl = plan.forward_flops(B)
        rec = dict(what='config_forward', config=name, ms_graph=ms, tflops=fl / ms / 1e9, gflop_per_img=fl / B / 1e9,
                   img_per_s_2000_steps=B / (2000 * ms * 1e-3), ws_gb=plan.workspace_bytes(B) / 1e9, ops=plan.num_ops(B),
                   finite=bool(torch.isfinite(out).all()))
        with open(out_path, 'a') as f:
            f.write(json.dumps(rec) + '\n')
        print(rec, flush=True)
        del netG, x, out


def train_time(B, out_path, iters=5, config='sr3_16_128'):
    """Training step (forward + backward + Adam) timing at batch B: SR3 16->128 (C3), DDPM-128 (C5) or SR3 64->512."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import model as Model
    opt = bench.config_opt(config, phase='train')
    opt['model']['unet']['dropout'] = 0
    S = bench.CONFIGS[config]['size']
    torch.manual_seed(0)
    m = Model.create_model(opt)
    d = torch.device('cuda:0')
    data = {'HR': torch.rand(B, 3, S, S) * 2 - 1, 'SR': torch.rand(B, 3, S, S) * 2 - 1}
    m.feed_data(data)
    for _ in range(2):
        m.optimize_parameters()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        m.optimize_parameters()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / iters
    fl = 3 * m.netG.denoise_fn.plan.forward_flops(B)
    rec = dict(what='train_step', config=config, B=B, ms=dt * 1e3, img_per_s=B / dt, tflops_3x_fwd=fl / dt / 1e12, l_pix=m.get_current_log()['l_pix'],
               ws_gb=m.netG.denoise_fn._train_ws.numel() / 1e9)
    with open(out_path, 'a') as f:
        f.write(json.dumps(rec) + '\n')
    print(rec, flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--sweep', action='store_true')
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--unet', action='store_true')
    ap.add_argument('--only', default='')
    ap.add_argument('--cfgs', default='1,2,3,4')
    ap.add_argument('--kss', default='1,2,4,8')
    ap.add_argument('--tag', default='')
    ap.add_argument('--split', default='')
    ap.add_argument('--configs', action='store_true')
    ap.add_argument('--opcmp', action='store_true')
    ap.add_argument('--io', action='store_true')
    ap.add_argument('--train', default='')
    ap.add_argument('--train-config', default='sr3_16_128')
    a = ap.parse_args()
    if a.train:
        for bb in [int(v) for v in a.train.split(',')]:
            train_time(bb, os.path.join(OUT, 'probe_train.jsonl'), config=a.train_config)
    if a.io:
        io_times(os.path.join(OUT, 'probe_io.jsonl'))
    if a.opcmp:
        op_compare(a.batch, os.path.join(OUT, 'probe_opcmp.txt'))
    if a.configs:
        config_times(os.path.join(OUT, 'probe_configs.jsonl'))
    if a.split:
        for ns in [int(v) for v in a.split.split(',')]:
            unet_time_split(a.batch, ns, os.path.join(OUT, 'probe_unet.jsonl'))
    if a.unet:
        for fuse in (0, 1):
            unet_time(a.batch, os.path.join(OUT, 'probe_unet.jsonl'), fuse)
        unet_time(a.batch, os.path.join(OUT, 'probe_unet.jsonl'), 1, split_bf16=1)
    if a.sweep:
        conv_sweep(a.batch, os.path.join(OUT, 'probe_conv_B%d%s.jsonl' % (a.batch, a.tag)), a.quick,
                   only=[x for x in a.only.split(',') if x] or None, cfgs=[int(x) for x in a.cfgs.split(',')],
                   kss=[int(x) for x in a.kss.split(',')])
