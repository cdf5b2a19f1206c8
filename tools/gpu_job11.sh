#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/j11; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "stress" 2>&1 | tail -8 > $O/pytest_stress.txt
timeout 300 python tools/op_table.py > $O/op_table.txt 2> $O/op_table.err
timeout 600 python bench.py --steps 400 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-other-configs > $O/bench_quick.json 2> $O/bench_quick.err
tail -3 $O/pytest_stress.txt; tail -12 $O/op_table.txt; cut -c1-400 $O/bench_quick.json
