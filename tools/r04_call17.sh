#!/bin/bash
# validation of the gemm_split default: network-level golden / option tests, training parity, C2 gate, 2000-step drift, quick bench
set -u
OUT=gpurun_out/r04s; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_train.py -q -x -m gpu -s > $OUT/pytest_unet_train.log 2>&1; echo "unet/train rc=$?"; tail -2 $OUT/pytest_unet_train.log
timeout 900 python -m pytest tests/test_gpu_bench_configs.py -q -x -m gpu -s -k "wino_split_gate or (c3_train and 8-uniform-1)" > $OUT/pytest_configs.log 2>&1; echo "configs rc=$?"; tail -2 $OUT/pytest_configs.log
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-split-leg --no-other-configs > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc=$?"; cut -c1-260 $OUT/bench_quick.json
timeout 900 python -m pytest tests/test_gpu_trajectory.py -q -x -m gpu -s -k "test_c2_sr3_16_128_batch16_full" > $OUT/pytest_trajectory_c2.log 2>&1; echo "trajectory rc=$?"; grep -E "steps:|CPU oracle|passed|failed" $OUT/pytest_trajectory_c2.log | cut -c1-400
