#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/j18; mkdir -p $O
for d in 0 1 2 4 6 7; do
  SR3_ATTN_DBG=$d timeout 300 python tools/op_table.py --reps 3 > $O/op_table_d$d.txt 2> $O/err_d$d.txt; echo "dbg $d: $(grep -E '^ *(49|106) +60 ' $O/op_table_d$d.txt | awk '{print $3}' | tr '\n' ' ')"
done
