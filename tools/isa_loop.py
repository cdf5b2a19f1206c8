#!/usr/bin/env python3
"""Static view of a kernel's innermost loops from hipcc -S output (no GPU needed): opcode classes per loop body, spill
instructions, register / scratch figures.  Usage:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only csrc/conv3x3_wino.hip -o /tmp/w.s
  tools/isa_loop.py /tmp/w.s 'k_conv3x3_winoILi0ELb0ELb0ELb1' [--dump]
Counts are STATIC (both sides of a wave-uniform branch are counted), so they bound the executed counts from above."""
import re
import sys
from collections import Counter

PACKED = ('v_pk_fma_f32', 'v_pk_add_f32', 'v_pk_mul_f32', 'v_dot2c_f32_bf16', 'v_dot2_f32_bf16')


def classify(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op in PACKED or op.startswith('v_dot2'): return 'valu_packed_or_dot2 (does not hide under an MFMA)'
    if op in ('v_exp_f32_e32', 'v_rcp_f32_e32', 'v_log_f32_e32', 'v_rsq_f32_e32', 'v_sqrt_f32_e32'): return 'valu_transcendental'
    if op.startswith('v_mov') or op.startswith('v_accvgpr'): return 'valu_move'
    if op.startswith('v_'): return 'valu_plain'
    if op.startswith('ds_read') or op.startswith('ds_load'): return 'lds_read'
    if op.startswith('ds_'): return 'lds_write'
    if op.startswith('global_load') or op.startswith('buffer_load'): return 'vmem_load'
    if op.startswith('global_store') or op.startswith('buffer_store'): return 'vmem_store'
    if op.startswith('scratch_'): return 'scratch'
    if op == 's_waitcnt': return 's_waitcnt'
    if op == 's_barrier': return 's_barrier'
    if op == 's_nop': return 's_nop'
    if op.startswith('s_cbranch') or op == 's_branch': return 'branch'
    if op.startswith('s_'): return 'salu'
    return 'other'


def main():
    path, pat = sys.argv[1], sys.argv[2]
    dump = '--dump' in sys.argv
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(pat) + r'\w*:', l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    k = lines[start:end]
    print('kernel', lines[start].split(':')[0], '-', len(k), 'lines')
    for l in lines[end:end + 80]:
        if re.search(r'NumVgprs|NumAgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize', l):
            print('  ' + l.strip().lstrip('; '))
        if l.startswith('.Lfunc_end') and l != lines[end]:
            break
    # innermost loop (largest Depth) blocks: contiguous range from the first to the last line tagged with that depth
    depths = [int(m.group(1)) for l in k for m in [re.search(r'Depth=(\d+)', l)] if m]
    if not depths:
        print('no loops')
        return
    d = max(depths)
    idx = [i for i, l in enumerate(k) if 'Depth=%d' % d in l]
    # extend to the backward branch that closes the loop
    # the loop's blocks are the ones tagged with this depth (their label / %bb comment carries it); the body runs from the
    # first tagged line to the end of the last tagged block
    first = idx[0]
    last = idx[-1] + 1
    while last < len(k) and not (k[last].startswith('.LBB') or k[last].startswith('; %bb.')):
        last += 1
    last -= 1
    header = 'blocks tagged Depth=%d' % d
    body = k[first:last + 1]
    ops = [l.split()[0] for l in body if l.startswith('\t') and not l.strip().startswith(';') and not l.strip().startswith('.')]
    cls = Counter(classify(o) for o in ops)
    print('innermost loop (depth %d, %s): %d instructions' % (d, header, len(ops)))
    for c, n in sorted(cls.items(), key=lambda t: -t[1]):
        print('  %-58s %4d' % (c, n))
    det = Counter(o for o in ops if classify(o) in ('valu_move', 'valu_packed_or_dot2 (does not hide under an MFMA)', 'scratch', 'valu_transcendental'))
    print('  detail:', dict(det))
    whole = Counter(classify(l.split()[0]) for l in k if l.startswith('\t') and not l.strip().startswith(';') and not l.strip().startswith('.'))
    print('whole kernel: scratch instructions %d, mfma %d' % (whole.get('scratch', 0), whole.get('mfma', 0)))
    if dump:
        print('\n'.join(body))


if __name__ == '__main__':
    main()
