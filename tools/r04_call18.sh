#!/bin/bash
# two-step-ahead prefetch in the im2col kernels: op parity, then A/B of the headline step with gemm_split on / off in one run
set -u
OUT=gpurun_out/r04t; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "test_conv or gemm_split or stats or block_conv or dropout" > $OUT/pytest_ops.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_ops.log
Q="--steps 100 --warmup 5 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-split-leg --no-other-configs --no-exact-leg"
timeout 300 python bench.py $Q > $OUT/bench_split.json 2> $OUT/bench_split.err; echo "bench rc=$?"; cut -c1-200 $OUT/bench_split.json
timeout 300 python bench.py $Q --plan-opt gemm_split=0 > $OUT/bench_nosplit.json 2> $OUT/bench_nosplit.err; echo "bench rc=$?"; cut -c1-200 $OUT/bench_nosplit.json
timeout 300 python tools/op_table.py > $OUT/op_table.txt 2> $OUT/op_table.err
