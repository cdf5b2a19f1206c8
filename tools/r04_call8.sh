#!/bin/bash
# round-4 call 8: the training plan on the SPLIT instantiation (block1 / Upsample / dropout convs, 3x3 data gradients): parity + time
set -u
OUT=gpurun_out/r04i
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_bench_configs.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py -q -s -k "train or dropout or dist or fullsize or adam" > $OUT/pytest_train.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest_train.log | cut -c1-300
grep -h "training step" $OUT/pytest_train.log | cut -c1-230
timeout 300 python tools/grad_probe.py --batch 64 --gamma uniform --data-seed 8 --kink-margin 1e-4 --top 3 --variant default --variant wino_split=0 --variant winograd=0 > $OUT/probe_train_split.txt 2>&1
grep -h "^engine\|de-kinked" $OUT/probe_train_split.txt | cut -c1-170
