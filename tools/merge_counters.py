#!/usr/bin/env python3
"""The counter summaries bench.py reads (profiles/<round>_sq_counters.json, _hbm_traffic.json) come from the default plan's passes;
the all-fp32-MFMA plan's kernels (its own passes: <round>_exact_fp32_*) are added under their own symbols -- a symbol names one
instantiation, so nothing is overwritten.   python tools/merge_counters.py profiles/r06"""
import json
import sys


def main():
    prefix = sys.argv[1]
    for kind in ('sq_counters', 'hbm_traffic'):
        base = json.load(open('%s_%s.json' % (prefix, kind)))
        extra = json.load(open('%s_exact_fp32_%s.json' % (prefix, kind)))
        added = [k for k in extra if k not in base]
        for k in added:
            base[k] = dict(extra[k], plan='exact_fp32 (wino_split = gemm_split = attn_split = 0)') if isinstance(extra[k], dict) else extra[k]
        json.dump(base, open('%s_%s.json' % (prefix, kind), 'w'), indent=1)
        print(kind, 'added', added)


if __name__ == '__main__':
    main()
