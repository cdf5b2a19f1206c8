#!/usr/bin/env python3
"""Print the ordered launch list of one UNet forward as the plan compiles it (host-only; no GPU needed):
  python tools/dump_plan.py [--batch 16] [--config sr3_16_128|sr3_64_512|ddpm_128] [--split-bf16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd'))
from sr3_hip import engine as E      # noqa: E402

CONFIGS = {'sr3_16_128': ('sr3', 6, 3, 64, 32, [1, 2, 4, 8, 8], [16], 2, 128),
           'sr3_64_512': ('sr3', 6, 3, 64, 16, [1, 2, 4, 8, 16], [], 1, 512),
           'ddpm_128': ('ddpm', 3, 3, 64, 32, [1, 1, 2, 2, 4, 4], [16], 2, 128)}
KIND = {10: 'embed + FiLM rows', 20: 'input conv (NCHW -> NHWC)', 30: 'GroupNorm statistics', 40: 'GroupNorm fold',
        50: 'conv', 60: 'attention', 70: 'output block (NHWC -> NCHW)'}
TILE = {1: 'im2col 128x128', 2: 'im2col 128x64', 3: 'im2col 64x64', 4: 'im2col 64x128', 5: 'halo 128x128', 6: 'halo 256x64',
        7: 'halo 128x128 split-bf16', 8: 'halo 256x64 split-bf16 (8 waves)', 9: 'halo 256x128 (8 waves)',
        10: 'halo 256x128 split-bf16 (8 waves)', 11: 'Winograd F(2x2,3x3) 64 tiles x 64 (8 waves)',
        12: 'Winograd F(2x2,3x3) 3 x bf16 split (8 waves)', 13: 'Winograd F(2x2,3x3) 3 x bf16 split (4 waves)',
        14: 'im2col 128x128 3 x bf16 split', 15: 'im2col 128x64 3 x bf16 split', 16: 'im2col 64x64 3 x bf16 split',
        17: 'im2col 64x128 3 x bf16 split', 18: 'im2col 128x128 3 x bf16 split, pre-split weights',
        19: 'im2col 128x64 3 x bf16 split, pre-split weights', 20: 'im2col 64x64 3 x bf16 split, pre-split weights',
        21: 'im2col 64x128 3 x bf16 split, pre-split weights'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--config', default='sr3_16_128', choices=sorted(CONFIGS))
    ap.add_argument('--split-bf16', action='store_true')
    a = ap.parse_args()
    plan = E.Plan(*CONFIGS[a.config])
    if a.split_bf16:
        plan.set_option('split_bf16', 1)
    ops = plan.op_list(a.batch)
    print('# %s, batch %d%s: %d launches (+ split-K reduces), %.2f GFLOP / image, workspace %.2f GB'
          % (a.config, a.batch, ', split_bf16' if a.split_bf16 else '', len(ops), plan.forward_flops(a.batch) / a.batch / 1e9,
             plan.workspace_bytes(a.batch) / 1e9))
    for i, o in enumerate(ops):
        if o['kind'] == 50:
            d = '%dx%d%s %4d -> %4d @ %3dx%-3d  %-34s ksplit %d%s%s  %7.2f GFLOP' % (
                o['ksize'], o['ksize'], ' s2' if o['stride'] == 2 else (' up' if o['upsample'] else '   '), o['cin'], o['cout'],
                o['h_out'], o['w_out'], TILE.get(o['tile_cfg'], str(o['tile_cfg'])), o['ksplit'],
                ' +1x1 res_conv(%d)' % o['fused_res_conv_cin'] if o['fused_res_conv_cin'] else '',
                ' +stats' if o['fused_output_stats'] else '', o['flops'] / 1e9)
        elif o['kind'] == 60:
            d = 'N = %d, d = %d  %7.2f GFLOP' % (o['h_out'], o['cin'], o['flops'] / 1e9)
        else:
            d = ''
        print('%3d  %-28s %s' % (i, KIND[o['kind']], d))


if __name__ == '__main__':
    main()
