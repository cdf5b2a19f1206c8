#!/bin/bash
set -u
OUT=gpurun_out/r04v; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--steps 150 --warmup 5 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-split-leg --no-other-configs --no-exact-leg --no-roofline"
timeout 300 python bench.py $Q > $OUT/bench_split.json 2> $OUT/bench_split.err; echo "bench rc=$?"; cut -c1-200 $OUT/bench_split.json
timeout 300 python bench.py $Q --plan-opt gemm_split=0 > $OUT/bench_nosplit.json 2> $OUT/bench_nosplit.err; echo "bench rc=$?"; cut -c1-200 $OUT/bench_nosplit.json
timeout 300 python bench.py $Q > $OUT/bench_split2.json 2> $OUT/bench_split2.err; echo "bench rc=$?"; cut -c1-200 $OUT/bench_split2.json
timeout 300 python tools/gpu_probe.py --train 64 > $OUT/train_probe.log 2>&1; grep -h "train_step" $OUT/train_probe.log | cut -c1-200
