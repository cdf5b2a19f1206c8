#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/j19; mkdir -p $O
timeout 2000 python -m pytest tests -q -m gpu --durations=25 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
