#!/bin/bash
# round-4 call 6: where the SPLIT Winograd instantiation's time goes (compile-time ablations)
set -u
OUT=gpurun_out/r04g
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=$PWD/image-super-resolution-via-iterative-refinement_amd/csrc
timeout 900 python tools/wino_ablate.py --lib $C/build_abl/libsr3_ablate.so --opt wino_split=1 --kind 555 --dbg 0,1,4,8,16,32,128,256,444,452,445 --tag r04g_ablate > $OUT/ablate_split.txt 2>&1
cat $OUT/ablate_split.txt
