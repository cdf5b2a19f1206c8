#!/bin/bash
# round 3, GPU call 1: MFMA/VALU overlap microbenchmark, Winograd ablations, data-parallel tests through the drop-in
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/j1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap.hip -o /tmp/mfma_overlap 2>/dev/null && timeout 120 /tmp/mfma_overlap | tee gpurun_out/j1/mfma_overlap.txt
timeout 900 python tools/wino_ablate.py --lib $PWD/image-super-resolution-via-iterative-refinement_amd/csrc/build_abl/libsr3_ablate.so --dbg 0,1,2,4,8,16,32,38,62 2>&1 | tee gpurun_out/j1/wino_ablate.txt
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -15 | tee gpurun_out/j1/pytest_dist.txt
