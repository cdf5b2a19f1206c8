#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/j4
CS=$PWD/image-super-resolution-via-iterative-refinement_amd/csrc
SR3_WINO_DBG=64 SR3_LIBRARY=$CS/build_abl/libsr3_ablate.so timeout 300 python tools/wino_phases.py 2>&1 | grep -v "^{" | tee gpurun_out/j4/phases2.txt
