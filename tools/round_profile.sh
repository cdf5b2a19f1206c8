#!/bin/bash
# One GPU-box pass that produces what profiles/ holds for a round: smoke, the default bench line (headline + other_configs
# + training + baselines), rocprofv3 kernel statistics of the sampling leg and of a training run, the counter passes
# (FETCH_SIZE, WRITE_SIZE, SQ set -- each in its own run, --kernel-trace only), the per-launch table of the forward and the
# stock PyTorch-ROCm kernel statistics AFTER a MIOpen warm-up run (find results cached: no naive_conv_* find kernels).
# Usage (through gpurun):   bash tools/round_profile.sh <tag> [--with-tests] [--quick]
set -u
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
if [ "${2:-}" = "--with-tests" ]; then
  python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -2 $OUT/pytest_gpu.log
fi
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
Q="--no-cpu-baseline --no-torch-baseline --train-steps 0 --no-split-leg --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --steps 200 --warmup 5 $Q > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
# counter passes (FETCH_SIZE, WRITE_SIZE, SQ set -- each its own run, --kernel-trace only; retried once: rocprofv3 --pmc has segfaulted
# right after HSA initialisation on some boxes of the pool), then the same for the all-fp32-MFMA plan
bash tools/pmc_passes.sh $TAG
bash tools/pmc_passes.sh ${TAG}_exact_fp32 --exact-fp32
python tools/op_table.py > $OUT/op_table.txt 2> $OUT/op_table.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -o train -- python tools/gpu_probe.py --train 64 > $OUT/train_probe.log 2>&1
# stock PyTorch-ROCm: first run fills MIOpen's user find-db, the profiled second run reuses it
python tools/torch_baseline_probe.py --config sr3_16_128 --batch 16 --steps 3 > $OUT/torch_warm.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/torch_stats -o torch -- python tools/torch_baseline_probe.py --config sr3_16_128 --batch 16 --steps 5 > $OUT/torch_probe.log 2>&1
if [ "${2:-}" != "--quick" ] && [ "${3:-}" != "--quick" ]; then
  # the multi-rank entry paths on this 1-GPU box: collective path forced on one rank and the launcher form the driver uses
  D="--steps 20 --warmup 3 --train-steps 2 --no-cpu-baseline --no-torch-baseline --no-roofline --no-other-configs"
  SR3_BENCH_FORCE_DIST=1 python bench.py --gpus 1 $D > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err; echo "force_dist rc=$?"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 $D > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err; echo "torchrun rc=$?"
fi
grep -h "train_step" $OUT/train_probe.log | cut -c1-260
for f in $OUT/bench.json; do cut -c1-400 $f; echo; done
# keep the merged-back payload small: the per-dispatch traces are not needed, the statistics and counter summaries are
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete
ls $OUT | head -40
