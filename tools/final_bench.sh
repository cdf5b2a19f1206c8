#!/bin/bash
# bench line + kernel statistics + per-launch table on the current tree (the cheap half of tools/round_profile.sh)
set -u
TAG=${1:-r04final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-260 $OUT/bench.json
Q="--no-cpu-baseline --no-torch-baseline --train-steps 0 --no-split-leg --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --steps 200 --warmup 5 $Q > $OUT/bench_under_rocprof.json 2> $OUT/stats.err; echo "stats rc=$?"
python tools/op_table.py > $OUT/op_table.txt 2> $OUT/op_table.err
find $OUT -name "*kernel_trace.csv" -delete
