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


if __name__ == '__main__':
    main()
