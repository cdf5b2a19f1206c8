#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/j7
CS=$PWD/image-super-resolution-via-iterative-refinement_amd/csrc
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd or conv" 2>&1 | tail -3 | tee gpurun_out/j7/pytest_ops.txt
timeout 600 python -m pytest tests/test_gpu_bench_configs.py -x -q -k "c2 or c5_batch32" 2>&1 | tail -3 | tee gpurun_out/j7/pytest_cfg.txt
SR3_WINO_DBG=64 SR3_LIBRARY=$CS/build_abl/libsr3_ablate.so timeout 300 python tools/wino_phases.py 2>&1 | grep -v "^{" | tee gpurun_out/j7/phases.txt
timeout 300 python tools/wino_ablate.py --dbg 0 --tag j7/ablate_persist 2>&1 | tee gpurun_out/j7/ablate_persist.txt
