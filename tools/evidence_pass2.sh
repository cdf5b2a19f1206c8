#!/bin/bash
# Round evidence, pass 2 of 2 (through gpurun), AFTER the counter summaries of pass 1 were copied into profiles/: the default bench
# line (reads them for roofline.traffic / sq_counters), the 1-rank launcher lines, and the whole -m gpu suite.
#   bash tools/evidence_pass2.sh r06
set -u
TAG=${1:-r06}
OUT=gpurun_out/${TAG}_final; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
D="--steps 20 --warmup 3 --train-steps 2 --no-cpu-baseline --no-torch-baseline --no-roofline --no-other-configs"
SR3_BENCH_FORCE_DIST=1 python bench.py --gpus 1 $D > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err; echo "force_dist rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 $D > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err; echo "torchrun rc=$?"
python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
