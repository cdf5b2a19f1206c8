#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/j21; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_unet.py tests/test_gpu_ops.py -x -q -k "without_clipping or p_sample" 2>&1 | tail -4 | tee $O/pytest_noclip.txt
timeout 1500 bash tools/round_profile.sh r03 --quick 2>&1 | tail -30
