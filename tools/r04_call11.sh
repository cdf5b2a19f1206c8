#!/bin/bash
# round-4: quick check of the four-wave kernel: op-level parity (gate cases, stress, fused statistics) + step time + per-layer table
set -u
OUT=gpurun_out/${1:-r04l}
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "4wave or split4w or (fused_output_stats and 13)" > $OUT/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -2 $OUT/pytest_ops.log | cut -c1-200
Q="--steps 50 --warmup 5 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-other-configs --no-roofline --no-exact-leg"
SR3_WINO4=1 timeout 300 python bench.py $Q > $OUT/bench_4wave.json 2> $OUT/bench_4wave.err
python -c "
import json; d=json.load(open('$OUT/bench_4wave.json')); print('4wave ms_per_step', d['ms_per_step'], d['value'], d['config'].get('output_finite'))"
timeout 300 python tools/op_table.py --opt wino4=1 > $OUT/op_table_4wave.txt 2> $OUT/op_table_4wave.err; grep "#  575\|#  465" $OUT/op_table_4wave.txt; grep "winograd 3xbf16 4w" $OUT/op_table_4wave.txt | awk '{print $3, $4, $7, $8, $10, $11, $12, $13}' | sort | uniq -c | sort -k2 -n | head -24
