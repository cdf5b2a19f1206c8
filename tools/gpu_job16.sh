#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/j16; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attention or conv_in or conv_out" 2>&1 | tail -6 > $O/pytest_ops.txt; tail -3 $O/pytest_ops.txt
timeout 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_bench_configs.py -x -q -k "not train" 2>&1 | tail -8 > $O/pytest_cfg.txt; tail -3 $O/pytest_cfg.txt
timeout 300 python tools/op_table.py > $O/op_table.txt 2> $O/op_table.err; tail -12 $O/op_table.txt; grep -E " (20|60|70) +[0-9.]+ us" $O/op_table.txt
timeout 600 python bench.py --steps 400 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-other-configs --no-roofline > $O/bench_quick.json 2> $O/bench_quick.err; cut -c1-260 $O/bench_quick.json
( time timeout 1200 python tools/train_error_probe.py --batch 64 --dropout 0.2 --top 8 --skip-f32 --variant winograd=0 ) > $O/err64_variants.log 2>&1
tail -16 $O/err64_variants.log | cut -c1-200
