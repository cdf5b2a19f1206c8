#!/bin/bash
set -u
OUT=gpurun_out/r04r; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/mfma_overlap_bf16.hip -o /tmp/mfma_overlap_bf16 && timeout 120 /tmp/mfma_overlap_bf16 > $OUT/mfma_overlap_bf16.txt 2>&1; echo "rc=$?"; cat $OUT/mfma_overlap_bf16.txt
