#!/bin/bash
# round-4 call 7: the whole GPU test suite on the tree with wino_split as the default plan option, smoke, quick bench
set -u
OUT=gpurun_out/r04h
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest_gpu.log | cut -c1-300
grep -h "batch 16, 2000 steps\|batch 32, 2000 steps\|batch 4, 2000 steps\|CPU oracle over" $OUT/pytest_gpu.log | cut -c1-420
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-torch-baseline --train-steps 3 > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04h/bench_quick.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'], d['dtype'], 'parity', d.get('parity_max_abs'))
e=d.get('exact_fp32',{}); print('exact_fp32', {k:e.get(k) for k in ('ms_per_step','images_per_s_per_gpu','eps_max_abs_diff_vs_headline_plan','error')})
r=d.get('roofline',{}); print('roofline', {k:r.get(k) for k in ('kernel','achieved','peak','frac','avg_launch_us','launches_per_forward','error')})
print('train', d.get('train',{}).get('ms_per_step'))
print('other', {k:(v.get('ms_per_step'), v.get('parity_max_abs')) for k,v in d.get('other_configs',{}).items()})
PY
