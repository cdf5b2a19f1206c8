#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/j12; mkdir -p $O
timeout 900 python tools/gpu_probe.py --sweep --only net1x1 --cfgs 1,2,3,4,12 --kss 1,2,4 --tag _net1x1 > $O/sweep.log 2>&1
cp gpurun_out/probe_conv_B16_net1x1.jsonl $O/ 2>/dev/null
tail -3 $O/sweep.log
