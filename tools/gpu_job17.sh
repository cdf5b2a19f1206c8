#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/j17; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" 2>&1 | tail -6 > $O/pytest_ops.txt; tail -2 $O/pytest_ops.txt
timeout 300 python tools/op_table.py > $O/op_table.txt 2> $O/op_table.err; grep -E " (60) +[0-9.]+ us" $O/op_table.txt; tail -11 $O/op_table.txt | head -3
SR3_ATTN_V1=1 timeout 300 python tools/op_table.py > $O/op_table_v1.txt 2> $O/op_table_v1.err; grep -E " (60) +[0-9.]+ us" $O/op_table_v1.txt
