#!/bin/bash
# im2col kernels: start skew between the co-resident workgroups of a CU (SR3_CONV_DBG >> 8, units of 256 cycles per workgroup slot)
set -u
OUT=gpurun_out/r04q; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for sk in 0 2 4 8 16 32; do
  SR3_CONV_DBG=$((sk*256)) timeout 300 python tools/gpu_probe.py --sweep --only k1_16_512_1536,k1_16_512_512,k1_16_1024_512,k1_128_192_64 --cfgs 3,14,15,16 --kss 1 --tag _sk$sk > $OUT/sweep_sk$sk.log 2>&1; echo "sk$sk rc=$?"
  cp gpurun_out/probe_conv_B16_sk$sk.jsonl $OUT/ 2>/dev/null
done
