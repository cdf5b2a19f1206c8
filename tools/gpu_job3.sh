#!/bin/bash
# round 3, GPU call 3: persistent Winograd kernel -- parity, persistent vs not, ablations of the new kernel
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/j3
CS=$PWD/image-super-resolution-via-iterative-refinement_amd/csrc
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd or conv" 2>&1 | tail -3 | tee gpurun_out/j3/pytest_ops.txt
timeout 600 python -m pytest tests/test_gpu_bench_configs.py -x -q -k "c2 or c5_batch32" 2>&1 | tail -3 | tee gpurun_out/j3/pytest_cfg.txt
timeout 300 python tools/wino_ablate.py --dbg 0 --tag j3/ablate_persist 2>&1 | tail -3 | tee gpurun_out/j3/ablate_persist.txt
SR3_WINO_NONPERSISTENT=1 timeout 300 python tools/wino_ablate.py --dbg 0 --tag j3/ablate_nonpersist 2>&1 | tail -3 | tee gpurun_out/j3/ablate_nonpersist.txt
timeout 900 python tools/wino_ablate.py --lib $CS/build_abl/libsr3_ablate.so --dbg 0,1,2,4,8,32,38,62 --tag j3/ablate_new 2>&1 | tee gpurun_out/j3/ablate_new.txt
