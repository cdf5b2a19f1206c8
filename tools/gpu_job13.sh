#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/j13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q --durations=8 2>&1 | tail -25 > $O/pytest_ops.txt
timeout 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_bench_configs.py -x -q --durations=8 -k "not train" 2>&1 | tail -25 > $O/pytest_cfg.txt
timeout 300 python tools/op_table.py > $O/op_table.txt 2> $O/op_table.err
SR3_SPLITK_TAIL=0 timeout 300 python tools/op_table.py > $O/op_table_notail.txt 2> $O/op_table_notail.err
timeout 600 python bench.py --steps 400 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-other-configs --no-roofline > $O/bench_quick.json 2> $O/bench_quick.err
tail -4 $O/pytest_ops.txt; tail -4 $O/pytest_cfg.txt; tail -12 $O/op_table.txt; tail -12 $O/op_table_notail.txt; cut -c1-300 $O/bench_quick.json
