#!/bin/bash
# round-4 call 4: the Winograd kernel's 3 x bf16 split instantiation (wino_split): op-level gates, network gates, timing
set -u
OUT=gpurun_out/r04d
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -s -k "winograd" > $OUT/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $OUT/pytest_ops.log
grep -h "ks[0-9].*split\|tile 12" $OUT/pytest_ops.log | cut -c1-200 | head -60
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_bench_configs.py -q -s -k "wino_split or c5_train" > $OUT/pytest_net.log 2>&1; echo "net rc=$?"; tail -3 $OUT/pytest_net.log
grep -h "C2 batch 16\|graph-replayed\|training step" $OUT/pytest_net.log | cut -c1-260
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-other-configs --no-roofline > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04d/bench_quick.json'))
print('fp32 ms_per_step', d['ms_per_step'], 'value', d['value']); print('wino_split', d.get('wino_split'))
PY
timeout 300 python tools/op_table.py --opt wino_split=1 > $OUT/op_table_split.txt 2> $OUT/op_table_split.err; tail -14 $OUT/op_table_split.txt; grep "winograd" $OUT/op_table_split.txt | head -30
timeout 900 python -m pytest tests/test_gpu_trajectory.py -q -s -k "wino_split" > $OUT/pytest_traj.log 2>&1; echo "traj rc=$?"; grep -h "steps:" $OUT/pytest_traj.log | cut -c1-400; tail -2 $OUT/pytest_traj.log
