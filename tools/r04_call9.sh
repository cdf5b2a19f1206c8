#!/bin/bash
# round-4 call 9: the four-wave split Winograd kernel (conv3x3_wino4.hip): op-level parity, step time, per-layer table
set -u
OUT=gpurun_out/r04j
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -s -x -k "4wave or split4w or (fused_output_stats and 13)" > $OUT/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $OUT/pytest_ops.log | cut -c1-200
grep -h "tile 13" $OUT/pytest_ops.log | cut -c1-170 | head -30
Q="--steps 50 --warmup 5 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-other-configs --no-roofline --no-exact-leg"
timeout 300 python bench.py $Q > $OUT/bench_8wave.json 2> $OUT/bench_8wave.err
SR3_WINO4=1 timeout 300 python bench.py $Q > $OUT/bench_4wave.json 2> $OUT/bench_4wave.err
python - <<'PY'
import json
for n in ('8wave','4wave'):
    try:
        d=json.load(open('gpurun_out/r04j/bench_%s.json'%n)); print(n, 'ms_per_step', d['ms_per_step'], 'value', d['value'], d['config'].get('output_finite'))
    except Exception as e: print(n, 'failed', e)
PY
tail -3 $OUT/bench_4wave.err
timeout 300 python tools/op_table.py --opt wino4=1 > $OUT/op_table_4wave.txt 2> $OUT/op_table_4wave.err; tail -14 $OUT/op_table_4wave.txt; grep "winograd" $OUT/op_table_4wave.txt | head -20
