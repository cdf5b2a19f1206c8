#!/bin/bash
# One GPU-box pass that produces everything profiles/ holds for a round: the full GPU test log, smoke, the default
# bench line, rocprofv3 kernel statistics of the sampling leg and of a training run, and the two HBM PMC passes
# (counters in their own runs, --kernel-trace only).  Usage (through gpurun): bash tools/round_profile.sh <tag>
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -2 $OUT/pytest_gpu.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
Q="--no-cpu-baseline --train-steps 0 --no-split-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --steps 200 --warmup 5 $Q > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- python bench.py --steps 4 --warmup 1 --no-roofline $Q > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- python bench.py --steps 4 --warmup 1 --no-roofline $Q > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -o train -- python tools/gpu_probe.py --train 64 > $OUT/train_probe.log 2>&1
python tools/gpu_probe.py --configs > $OUT/configs.log 2>&1
python tools/gpu_probe.py --train 32 --train-config ddpm_128 >> $OUT/train_probe.log 2>&1
python tools/gpu_probe.py --train 2 --train-config sr3_64_512 >> $OUT/train_probe.log 2>&1
grep -h "train_step\|config_forward" $OUT/train_probe.log $OUT/configs.log | cut -c1-260
# keep the merged-back payload small: the per-dispatch traces are not needed, the statistics are
find $OUT -name "*kernel_trace.csv" -size +8M -delete
ls -la $OUT | head -30
