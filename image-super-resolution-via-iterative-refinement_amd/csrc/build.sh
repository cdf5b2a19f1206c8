#!/bin/bash
# Build libsr3_mi355x.so (gfx950 only) in-tree.  Usage: csrc/build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=../sr3_hip/libsr3_mi355x.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p build
pids=()
for f in conv_igemm conv3x3_halo conv3x3_wino gemm1x1 small_kernels attention plan train_kernels train_small wgrad attention_bwd train_plan io_metrics resize; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ sr3_common.h -nt build/$f.o ] || [ train.h -nt build/$f.o ] || [ plan_internal.h -nt build/$f.o ] || [ ../../include/sr3_mi355x.h -nt build/$f.o ] || [ ../../include/sr3_io_mi355x.h -nt build/$f.o ]; then
    $HIPCC $FLAGS "$@" -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/conv_igemm.o build/conv3x3_halo.o build/conv3x3_wino.o build/gemm1x1.o build/small_kernels.o build/attention.o build/plan.o build/train_kernels.o build/train_small.o build/wgrad.o build/attention_bwd.o build/train_plan.o build/io_metrics.o build/resize.o -o $OUT
echo "built $OUT"
