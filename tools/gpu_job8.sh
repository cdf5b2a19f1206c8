#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/j8
CS=$PWD/image-super-resolution-via-iterative-refinement_amd/csrc
for v in noslp prio noslp_prio; do
  timeout 300 python tools/wino_ablate.py --lib $CS/build_abl/libsr3_$v.so --dbg 0 --tag j8/ablate_$v 2>&1 | tail -2 | tee gpurun_out/j8/ablate_$v.txt
done
timeout 300 python tools/wino_ablate.py --dbg 0 --tag j8/ablate_main 2>&1 | tail -2 | tee gpurun_out/j8/ablate_main.txt
timeout 1500 python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_unet.py -x -q 2>&1 | tail -5 | tee gpurun_out/j8/pytest.txt
