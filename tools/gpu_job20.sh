#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/j20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "test_conv" 2>&1 | tail -8 > $O/pytest_ops.txt; tail -3 $O/pytest_ops.txt
timeout 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_bench_configs.py tests/test_gpu_train.py -x -q -k "not c3_train" 2>&1 | tail -8 > $O/pytest_cfg.txt; tail -3 $O/pytest_cfg.txt
for t in 0 22 21 11; do
  SR3_FRAG_TILE=$t timeout 300 python tools/op_table.py --reps 3 > $O/op_table_t$t.txt 2> $O/err_t$t.txt; echo "tile $t: $(tail -12 $O/op_table_t$t.txt | grep -E '# +(53|655|59) ' | tr '\n' ' ')"
done
SR3_NO_FRAG=1 timeout 300 python tools/op_table.py --reps 3 > $O/op_table_nofrag.txt 2> $O/err_nofrag.txt; echo "im2col: $(tail -12 $O/op_table_nofrag.txt | grep -E '# +(52|53|655|59) ' | tr '\n' ' ')"
timeout 600 python bench.py --steps 400 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-other-configs --no-roofline > $O/bench_quick.json 2> $O/bench_quick.err; cut -c1-260 $O/bench_quick.json
