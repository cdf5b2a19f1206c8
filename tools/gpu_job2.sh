#!/bin/bash
# round 3, GPU call 2: new Winograd kernel -- parity first, then A/B against the round-2 kernel and a no-SLP build
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/j2
CS=$PWD/image-super-resolution-via-iterative-refinement_amd/csrc
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd or conv" 2>&1 | tail -8 | tee gpurun_out/j2/pytest_ops.txt
timeout 900 python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_unet.py tests/test_gpu_dist.py -x -q 2>&1 | tail -8 | tee gpurun_out/j2/pytest_cfg.txt
for v in base noslp; do
  timeout 300 python tools/wino_ablate.py --lib $CS/build_abl/libsr3_$v.so --dbg 0 --tag j2/ablate_$v 2>&1 | tee gpurun_out/j2/ablate_$v.txt
done
timeout 300 python tools/wino_ablate.py --dbg 0 --tag j2/ablate_new 2>&1 | tee gpurun_out/j2/ablate_new.txt
