#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/j14; mkdir -p $O
T="tests/test_gpu_bench_configs.py::test_c2_batch16_forward_and_graph_step"
for m in 0 2 1; do
  SR3_TAIL_PLAN_SYNC=$m timeout 300 python -m pytest $T -x -q 2>&1 | grep -E "passed|failed|max abs err" | head -4 > $O/graph_mode$m.txt; echo "mode $m: $(tr '\n' ' ' < $O/graph_mode$m.txt)"
done
CS=$PWD/image-super-resolution-via-iterative-refinement_amd/csrc
SR3_LIBRARY=$CS/build_nf/libsr3_nofence.so timeout 300 python tools/op_table.py > $O/op_table_nofence.txt 2> $O/op_table_nofence.err
tail -12 $O/op_table_nofence.txt
grep -E "^ *(33|38|54|74|76|91) " $O/op_table_nofence.txt
