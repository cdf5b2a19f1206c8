#!/bin/bash
# where the im2col kernels' time goes: SR3_CONV_DBG ablations on four layer shapes, fp32 64x64 tile and the split tiles
set -u
OUT=gpurun_out/r04p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for dbg in 0 1 2 4 8 3; do
  SR3_CONV_DBG=$dbg timeout 300 python tools/gpu_probe.py --sweep --only k1_16_512_1536,k1_16_512_512,k1_16_1024_512,k1_128_192_64 --cfgs 3,14,15,16 --kss 1 --tag _dbg$dbg > $OUT/sweep_dbg$dbg.log 2>&1; echo "dbg$dbg rc=$?"
  cp gpurun_out/probe_conv_B16_dbg$dbg.jsonl $OUT/ 2>/dev/null
done
