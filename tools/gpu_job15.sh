#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/j15; mkdir -p $O
nproc > $O/nproc.txt
( time timeout 1200 python tools/train_error_probe.py --batch 64 --dropout 0.2 --top 16 ) > $O/err64.log 2>&1
tail -40 $O/err64.log
