#!/bin/bash
# round-4 call 10: where the four-wave split Winograd kernel's time goes (ablations) + its step time after the residual prefetch
set -u
OUT=gpurun_out/r04k
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=$PWD/image-super-resolution-via-iterative-refinement_amd/csrc
timeout 900 python tools/wino_ablate.py --lib $C/build_abl/libsr3_ablate.so --opt wino_split=1,wino4=1 --kind 575 --dbg 0,1,4,8,16,32,128,180,181 --tag r04k_ablate > $OUT/ablate_wino4.txt 2>&1
cat $OUT/ablate_wino4.txt
Q="--steps 50 --warmup 5 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-other-configs --no-roofline --no-exact-leg"
SR3_WINO4=1 timeout 300 python bench.py $Q > $OUT/bench_4wave.json 2> $OUT/bench_4wave.err
python -c "
import json; d=json.load(open('gpurun_out/r04k/bench_4wave.json')); print('4wave ms_per_step', d['ms_per_step'], d['value'])"
