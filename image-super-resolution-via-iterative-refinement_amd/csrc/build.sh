#!/bin/bash
# Build libsr3_mi355x.so (gfx950 only) in-tree.  Usage: csrc/build.sh [--clean] [extra hipcc flags]
#   --clean            drop the object cache first, so every source goes through the compiler
#   SR3_BUILD_DIR=dir  object directory (default build/); SR3_OUT=path the library to write (default ../sr3_hip/libsr3_mi355x.so)
#                      -- A/B builds: SR3_BUILD_DIR=build_x SR3_OUT=/path/libx.so csrc/build.sh -DFOO, then SR3_LIBRARY=/path/libx.so
set -e
cd "$(dirname "$0")"
OUT=${SR3_OUT:-../sr3_hip/libsr3_mi355x.so}
BUILD=${SR3_BUILD_DIR:-build}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
if [ "${1:-}" = "--clean" ]; then rm -rf "$BUILD"; shift; fi
mkdir -p "$BUILD"
SRCS="conv_igemm gemm1x1 conv3x3_halo conv3x3_wino conv3x3_wino2 small_kernels attention plan train_kernels train_small wgrad attention_bwd train_plan io_metrics resize"
pids=()
for f in $SRCS; do
  o=$BUILD/$f.o
  if [ ! -f $o ] || [ $f.hip -nt $o ] || [ sr3_common.h -nt $o ] || [ train.h -nt $o ] || [ plan_internal.h -nt $o ] || [ ../../include/sr3_mi355x.h -nt $o ] || [ ../../include/sr3_io_mi355x.h -nt $o ]; then
    $HIPCC $FLAGS "$@" -c $f.hip -o $o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for f in $SRCS; do OBJS="$OBJS $BUILD/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT
echo "built $OUT"
