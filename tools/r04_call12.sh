#!/bin/bash
set -u
OUT=gpurun_out/r04n
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=$PWD/image-super-resolution-via-iterative-refinement_amd/csrc
timeout 900 python tools/wino_ablate.py --lib $C/build_abl/libsr3_ablate.so --opt wino_split=1,wino4=1 --kind 575 --dbg 0,32,64,80,180 --tag r04n_ablate > $OUT/ablate_wino4.txt 2>&1
cat $OUT/ablate_wino4.txt | cut -c1-150
