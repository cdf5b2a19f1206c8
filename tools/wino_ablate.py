#!/usr/bin/env python3
"""Where the Winograd kernel's time goes: per-layer HIP-event times of one SR3 16->128 forward (batch 16, the plan the
bench runs) for the production kernel and for its compile-time ablations (conv3x3_wino.hip, template DBG; library built
with -DSR3_WINO_ABLATIONS).  One child process per variant (the knob is read once per process).  Usage on the GPU box:
    python tools/wino_ablate.py --lib /path/to/libsr3_ablate.so [--dbg 0,1,2,...] [--config sr3_16_128]
Writes gpurun_out/wino_ablate.json and prints one table row per distinct layer shape."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd')


def child(cfg_name, reps):
    sys.path.insert(0, PKG)
    sys.path.insert(0, ROOT)
    import torch
    import bench
    import model.networks as networks
    from sr3_hip import lib as L
    cfg = bench.CONFIGS[cfg_name]
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    netG = networks.define_G(bench.config_opt(cfg_name)).to(dev)
    un = netG.denoise_fn
    plan = un.plan
    plan.set_option('fuse_stats', 1)
    for kv in os.environ.get('SR3_ABLATE_OPTS', '').split(','):
        if kv:
            plan.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    un.ensure_derived()
    B, S = cfg['batch'], cfg['size']
    x = torch.randn(B, 3, S, S, device=dev)
    cond = (torch.rand(B, 3, S, S, device=dev) * 2 - 1) if cfg['conditional'] else None
    lib = L.load()
    if plan.variant == 'sr3':
        level, tstep = torch.full((B,), 0.5, device=dev), None
    else:
        level, tstep = None, torch.full((B,), 1000, dtype=torch.long, device=dev)
    wsbuf, need = un._ws.get(plan, B, dev)
    out = torch.empty(B, 3, S, S, device=dev)
    max_ops = 4096
    ms = (C.c_float * max_ops)()
    kind = (C.c_int * max_ops)()
    fl = (C.c_double * max_ops)()
    n = C.c_int()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    acc = None
    for r in range(reps + 1):
        L.check(lib.sr3_unet_forward_profile(plan.handle, L.ptr(x), L.ptr(cond), 0 if cond is None else 3, L.ptr(level),
                                             L.ptr(tstep), L.ptr(un.freq), L.ptr(un.arena.data), L.ptr(wsbuf), need, L.ptr(out),
                                             B, stream, max_ops, ms, kind, fl, C.byref(n)))
        if r == 0:
            continue
        if acc is None:
            acc = [0.0] * n.value
        for i in range(n.value):
            acc[i] += ms[i] / reps
    ops = plan.op_list(B)
    rows = []
    j = -1
    # sr3_unet_forward_profile reports one entry per plan op, plus a kind-59 entry after an op that ran a split-K reduce
    for i in range(n.value):
        if int(kind[i]) != 59:
            j += 1
        o = ops[j]
        label = '%dx%d %d->%d%s ks%d' % (o['h_out'], o['w_out'], o['cin'], o['cout'], ' up' if o['upsample'] else '', o['ksplit'])
        rows.append((int(kind[i]), acc[i], float(fl[i]), label))
    assert j == len(ops) - 1, (j, len(ops))
    print('ABLATE ' + json.dumps(dict(rows=rows, total_ms=sum(acc))), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default=None)
    ap.add_argument('--dbg', default='0,1,2,4,8,16,32,38,46,62')
    ap.add_argument('--config', default='sr3_16_128')
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--child', action='store_true')
    ap.add_argument('--tag', default='wino_ablate')
    ap.add_argument('--opt', default='', help='plan options for the child, e.g. wino_split=1')
    ap.add_argument('--kind', type=int, default=455, help='profile kind to tabulate (455 fp32 Winograd, 555 its 3 x bf16 split form)')
    a = ap.parse_args()
    if a.child:
        return child(a.config, a.reps)
    res = {}
    for d in [int(v) for v in a.dbg.split(',')]:
        env = dict(os.environ, SR3_WINO_DBG=str(d), SR3_ABLATE_OPTS=a.opt)
        if a.lib:
            env['SR3_LIBRARY'] = a.lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', '--config', a.config, '--reps', str(a.reps)],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        line = [l for l in r.stdout.decode().splitlines() if l.startswith('ABLATE ')]
        if r.returncode != 0 or not line:
            print('dbg %d failed: %s' % (d, r.stderr.decode()[-400:]))
            continue
        res[d] = json.loads(line[0][7:])
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', a.tag + '.json'), 'w') as f:
        json.dump(res, f)
    # table: Winograd launches (kind 455) grouped by flops (= layer shape), ms per launch for each variant
    if 0 not in res:
        return
    base = res[0]['rows']
    groups = {}
    for i, (k, ms, fl, label) in enumerate(base):
        if k == a.kind:
            groups.setdefault(label, []).append(i)
    ds = sorted(res)
    print('us per launch      %-26s %3s %6s ' % ('layer', 'n', 'GFLOP') + ' '.join('%7s' % ('dbg%d' % d) for d in ds))
    for label, idx in sorted(groups.items(), key=lambda kv: -int(kv[0].split('x')[0])):
        cells = []
        for d in ds:
            rows = res[d]['rows']
            cells.append('%7.1f' % (1e3 * sum(rows[i][1] for i in idx) / len(idx)))
        print('                   %-26s %3d %6.2f ' % (label, len(idx), base[idx[0]][2] / 1e9) + ' '.join(cells))
    print('%-56s' % 'Winograd launches, ms per forward' + ' '.join('%7.3f' % sum(r[1] for r in res[d]['rows'] if r[0] == a.kind) for d in ds))
    print('%-56s' % 'forward, ms' + ' '.join('%7.3f' % res[d]['total_ms'] for d in ds))


if __name__ == '__main__':
    main()
