#!/bin/bash
# round-4 call 5: scheduling experiments on the SPLIT Winograd instantiation (A/B libraries)
set -u
OUT=gpurun_out/r04f
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=$PWD/image-super-resolution-via-iterative-refinement_amd/csrc
Q="--steps 50 --warmup 5 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-other-configs --no-roofline"
for n in base rot0 rot2 rot1prio; do
  if [ $n = base ]; then L=""; else L=$C/build_$n/libsr3_$n.so; fi
  SR3_LIBRARY=$L timeout 300 python bench.py $Q > $OUT/bench_$n.json 2> $OUT/bench_$n.err
  python - $n <<'PY'
import json,sys
n=sys.argv[1]
d=json.load(open('gpurun_out/r04f/bench_%s.json'%n))
w=d.get('wino_split',{})
print('%-7s fp32 %.3f ms   wino_split %s ms   diff vs fp32 %s' % (n, d['ms_per_step'], w.get('ms_per_step'), w.get('eps_max_abs_diff_vs_exact_fp32')))
PY
done
