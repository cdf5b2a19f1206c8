#!/usr/bin/env python3
"""Register / scratch / LDS record of every kernel of libsr3_mi355x.so: compiles each csrc/*.hip for gfx950 with -save-temps into a
scratch directory (the same flags as csrc/build.sh) and reads the .amdhsa_ kernel descriptors and the scratch_ instructions of the
generated assembly.  CPU only (hipcc cross-compiles).   python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd', 'csrc')
FLAGS = '--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -save-temps'.split()


def demangle(names):
    p = subprocess.run(['c++filt'], input='\n'.join(names).encode(), stdout=subprocess.PIPE)
    return p.stdout.decode().split('\n')


def main():
    tmp = tempfile.mkdtemp(prefix='sr3_res_')
    rows = []
    for src in sorted(glob.glob(os.path.join(CSRC, '*.hip'))):
        base = os.path.basename(src)[:-4]
        subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-c', src, '-o', os.path.join(tmp, base + '.o')], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, base + '-hip-amdgcn-amd-amdhsa-gfx950.s')).read()
        for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', asm, re.S):
            name, desc = m.group(1), m.group(2)
            g = lambda k: int(re.search(r'\.amdhsa_' + k + r'\s+(\S+)', desc).group(1))
            # the whole function body up to its .Lfunc_end label (NOT to the first s_endpgm: a kernel with an early exit has several,
            # and round 5's table reported 0 MFMA instructions for every k_conv_igemm<64,64,...> instantiation because of it)
            body = re.search(r'^' + re.escape(name) + r':.*?^\.Lfunc_end\d+:', asm, re.S | re.M).group(0).split('\n')
            mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
            sc = [i for i, l in enumerate(body) if re.match(r'\s*scratch_(load|store)', l)]
            inloop = sum(1 for i in sc if mf and mf[0] < i < mf[-1])
            vg, acc = g('next_free_vgpr'), g('accum_offset')
            rows.append((base, name, vg, max(0, vg - acc), g('next_free_sgpr'), g('private_segment_fixed_size'), len(sc), inloop,
                         g('group_segment_fixed_size'), len(mf)))
    names = demangle([r[1] for r in rows])
    print('file              vgpr+agpr  agpr  sgpr  scratch B  scratch instr (between first / last MFMA)  static LDS B  MFMA instr  kernel')
    for r, n in zip(rows, names):
        n = re.sub(r'\(.*', '', n).replace('void ', '').replace('sr3::', '')
        print('%-17s %9d %5d %5d %10d %14d (%d) %25d %11d  %s' % (r[0], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], n))
    print('\nvgpr+agpr = .amdhsa_next_free_vgpr (unified file: 512 per SIMD lane; <= 256 => two waves per SIMD, <= 128 => four); dynamic LDS is'
          ' set at launch\n(ensure_max_lds) and not part of the descriptor.  A kernel whose scratch instructions all lie outside the MFMA range spills only in its'
          ' prologue / epilogue.')


if __name__ == '__main__':
    main()
