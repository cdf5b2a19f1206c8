#!/usr/bin/env python3
"""Aggregate one rocprofv3 SQ-counter pass (its own run, --kernel-trace only) into per-kernel averages and the derived
MFMA-pipe-busy fraction:

  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \\
            SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -o bench -- python bench.py ...
  python tools/pmc_sq.py <dir> profiles/r02

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)   (MI355X_MICROARCH.md, rocprofv3 PMC slots:
SQ counters are summed over all SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs); wait_inst = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
(issue stalls), wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked at s_waitcnt / barrier).
Writes <prefix>_sq_counters.json (read by bench.py for roofline.sq_counters) and <prefix>_sq_counters.csv."""
import csv
import glob
import json
import os
import re
import sys


def main():
    d, prefix = sys.argv[1:3]
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit('no *counter_collection.csv under ' + d)
    agg = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            name = re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '').strip()
            if not name.startswith('sr3::'):
                continue
            k = agg.setdefault(name, {})
            c = k.setdefault(r['Counter_Name'], [0, 0.0])
            c[0] += 1
            c[1] += float(r['Counter_Value'])
    out = {}
    for name, cs in agg.items():
        avg = {c: v[1] / v[0] for c, v in cs.items()}
        n = max(v[0] for v in cs.values())
        rec = dict(dispatches=n, **{c: avg[c] for c in sorted(avg)})
        gui = avg.get('GRBM_GUI_ACTIVE')
        if gui and 'SQ_VALU_MFMA_BUSY_CYCLES' in avg:
            rec['mfma_busy'] = avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * gui / 8.0)
            rec['kernel_cycles_per_xcd'] = gui / 8.0
        if avg.get('SQ_WAVE_CYCLES'):
            for k2, c in (('wait_inst', 'SQ_WAIT_INST_ANY'), ('wait_any', 'SQ_WAIT_ANY'), ('active_inst', 'SQ_ACTIVE_INST_ANY')):
                if c in avg:
                    rec[k2] = avg[c] / avg['SQ_WAVE_CYCLES']
        out[name] = rec
    att = [v for k, v in out.items() if 'k_attention' in k and 'bwd' not in k]     # k_attention<..> / k_attention_v2<..>
    if att:
        out['attention'] = max(att, key=lambda v: v['dispatches'])
    json.dump(out, open(prefix + '_sq_counters.json', 'w'), indent=1)
    cols = ['kernel', 'dispatches', 'mfma_busy', 'wait_inst', 'wait_any', 'active_inst', 'SQ_INSTS_VALU', 'SQ_VALU_MFMA_BUSY_CYCLES',
            'SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE']
    with open(prefix + '_sq_counters.csv', 'w') as f:
        w = csv.writer(f)
        w.writerow(cols)
        for name, r in sorted(out.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) * kv[1]['dispatches']):
            if name == 'attention':
                continue
            w.writerow([name] + [('%.4g' % r[c] if isinstance(r.get(c), float) else r.get(c, '')) for c in cols[1:]])
    print('wrote', prefix + '_sq_counters.json', len(out), 'kernels')


if __name__ == '__main__':
    main()
