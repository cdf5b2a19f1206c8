#!/bin/bash
# split im2col kernel: parity + gate tests, then the tile sweep over every 1x1 / stride-2 shape of the C2 forward
set -u
OUT=gpurun_out/r04o; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "test_conv or gemm_split" -s > $OUT/pytest_ops.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_ops.log
timeout 600 python tools/gpu_probe.py --sweep --only net1x1 --cfgs 3,14,15,16,17 --kss 1,2,4 --tag _split > $OUT/sweep.log 2>&1; echo "sweep rc=$?"
cp gpurun_out/probe_conv_B16_split.jsonl $OUT/ 2>/dev/null
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-split-leg --no-other-configs --no-exact-leg > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench_quick.json
