#!/usr/bin/env python3
"""Aggregate two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs as
MI355X_MICROARCH.md prescribes) into per-kernel HBM traffic per launch.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/fetch -o bench -- python bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/write -o bench -- python bench.py ...
  python tools/pmc_traffic.py gpurun_out/prof/fetch gpurun_out/prof/write profiles/r01

Corrections (the guide's HBM / rocprofv3 section): both counters are in KB; on gfx950 FETCH_SIZE under-reports wide
coalesced reads by 2x, so reads = 2 * FETCH_SIZE.  Writes <prefix>_hbm_traffic.json (read by bench.py for
`roofline.traffic`) and <prefix>_bench_hbm_pmc.csv."""
import csv
import glob
import json
import os
import re
import sys


def load(d, counter):
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit('no *counter_collection.csv under ' + d)
    agg = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            name = re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '').strip()
            a = agg.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    return agg


def load_seq(d, counter, kernel_substr):
    """Counter values of one kernel in dispatch order."""
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and kernel_substr in r['Kernel_Name']:
                rows.append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    return [v for _, v in sorted(rows)]


def by_layer(fetch_dir, write_dir, prefix, kernel_substr='k_conv3x3_wino<0, false, false, true>', tile=12, config='sr3_16_128'):
    """Per-layer-class HBM traffic of the dominant kernel: the graph replays the forward's launches in plan order, so the i-th
    launch of the kernel inside a step is the i-th op of the plan's launch list that runs it.  Host-only plan inspection."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'image-super-resolution-via-iterative-refinement_amd'))
    import bench
    from sr3_hip import engine as E
    c = bench.CONFIGS[config]
    u = c['unet']
    plan = E.Plan(c['which'], u['in_channel'], u['out_channel'], u['inner_channel'], u.get('norm_groups', 32), u['channel_multiplier'],
                  u['attn_res'], u['res_blocks'], c['size'])
    B = c['batch']
    ops = [o for o in plan.op_list(B) if o['kind'] == 50 and o['tile_cfg'] == tile and o['h_out'] >= 16]
    n = len(ops)
    fe, wr = load_seq(fetch_dir, 'FETCH_SIZE', kernel_substr), load_seq(write_dir, 'WRITE_SIZE', kernel_substr)
    if not n or len(fe) < n or len(wr) < n:
        print('by-layer: not enough dispatches of', kernel_substr, len(fe), len(wr), n)
        return
    fe, wr = fe[len(fe) % n:], wr[len(wr) % n:]           # whole forwards, counted from the end
    classes = {}
    for i, o in enumerate(ops):
        f = sum(fe[i::n]) / len(fe[i::n])
        w = sum(wr[i::n]) / len(wr[i::n])
        px = B * o['h_out'] * o['w_out']
        alg = 4.0 * (px // (4 if o['upsample'] else 1) * o['cin'] + px * o['cout']) + 6.0 * 16.0 * o['cin'] * o['cout']
        key = (o['h_out'], o['cin'], o['cout'], o['upsample'], o['ksplit'])
        a = classes.setdefault(key, [0, 0.0, 0.0, alg])
        a[0] += 1; a[1] += 2 * f * 1024; a[2] += w * 1024
    with open(prefix + '_wino_traffic_by_layer.txt', 'w') as out:
        out.write('# HBM traffic of %s per launch by layer class (%s, batch %d): reads = 2 x FETCH_SIZE, writes = WRITE_SIZE (KB counters,\n'
                  '# separate rocprofv3 --pmc passes); algorithmic = input + output tensors once (fp32) + the split filters once (16 positions x\n'
                  '# 3 bf16 planes = 6 bytes per (position, cin, cout)); split-K layers also write / the reduce kernel re-reads their slabs\n'
                  % (kernel_substr, config, B))
        out.write('# map  cin->cout  up ks  launches   read MB  write MB  total MB  algorithmic MB  ratio\n')
        tot = [0.0, 0.0]
        for key in sorted(classes, key=lambda k: (-k[0], k[1], k[2])):
            cnt, r, w, alg = classes[key]
            r, w = r / cnt, w / cnt
            tot[0] += (r + w) * cnt; tot[1] += alg * cnt
            out.write('%4d  %4d->%-4d  %d  %d  %6d   %8.1f  %8.1f  %8.1f  %8.1f   %5.2f\n'
                      % (key[0], key[1], key[2], key[3], key[4], cnt, r / 1e6, w / 1e6, (r + w) / 1e6, alg / 1e6, (r + w) / alg))
        out.write('# all %d launches: %.2f GB per forward against %.2f GB algorithmic: %.2fx\n' % (n, tot[0] / 1e9, tot[1] / 1e9, tot[0] / tot[1]))
    print('wrote', prefix + '_wino_traffic_by_layer.txt')


def main():
    fetch_dir, write_dir, prefix = sys.argv[1:4]
    fe, wr = load(fetch_dir, 'FETCH_SIZE'), load(write_dir, 'WRITE_SIZE')
    out = {}
    rows = []
    for name in sorted(set(fe) | set(wr), key=lambda n: -(2 * fe.get(n, [0, 0])[1] + wr.get(n, [0, 0])[1])):
        nf, sf = fe.get(name, [0, 0.0])
        nw, sw = wr.get(name, [0, 0.0])
        n = max(nf, nw)
        if n == 0 or not name.startswith('sr3::'):
            continue
        fkb, wkb = (sf / nf if nf else 0.0), (sw / nw if nw else 0.0)
        out[name] = dict(launches=n, fetch_kb=fkb, write_kb=wkb, hbm_bytes_per_launch=(2 * fkb + wkb) * 1024)
        rows.append((name, n, round(fkb), round(wkb), round(2 * fkb / 1024, 1), round(wkb / 1024, 1), round((2 * fkb + wkb) / 1024, 1)))
    json.dump(out, open(prefix + '_hbm_traffic.json', 'w'), indent=1)
    with open(prefix + '_bench_hbm_pmc.csv', 'w') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'launches', 'FETCH_SIZE_KB_avg', 'WRITE_SIZE_KB_avg', 'hbm_read_MB_corrected_x2', 'hbm_write_MB', 'hbm_total_MB_per_launch'])
        w.writerows(rows)
    print('wrote', prefix + '_hbm_traffic.json', len(out), 'kernels')
    try:
        by_layer(fetch_dir, write_dir, prefix)
    except Exception as e:                  # the per-kernel summary above is what bench.py needs; this table is extra
        print('by-layer table failed: %s: %s' % (type(e).__name__, e))


if __name__ == '__main__':
    main()
