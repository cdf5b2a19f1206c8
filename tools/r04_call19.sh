#!/bin/bash
# im2col tiles IN the forward (cold operands): per-launch table for every tile, split and fp32
set -u
OUT=gpurun_out/r04u; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for sp in 1 0; do for t in 1 2 3 4; do
  timeout 200 python tools/op_table.py --opt gemm_split=$sp --opt gemm_tile=$t > $OUT/op_table_split${sp}_tile$t.txt 2> $OUT/err_${sp}_$t.txt; echo "split$sp tile$t rc=$?"
done; done
