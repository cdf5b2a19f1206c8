#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/j23; mkdir -p $O
D="--steps 20 --warmup 3 --train-steps 2 --no-cpu-baseline --no-torch-baseline --no-roofline --no-other-configs"
SR3_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 $D > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo "force_dist rc=$?"; cut -c1-200 $O/bench_force_dist.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 $D > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "torchrun rc=$?"; cut -c1-200 $O/bench_torchrun1.json
timeout 120 python bench.py --gpus 2 $D > $O/bench_gpus2.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$? (2 = refused, expected on a 1-GPU box)"; tail -2 $O/bench_gpus2.err
