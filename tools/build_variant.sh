#!/bin/bash
# A/B builds of ONE csrc source: tools/build_variant.sh <name> <source without .hip> [extra hipcc flags]
#   -> tools/bin/libsr3_<name>.so = the default library's objects (csrc/build/) with that one source recompiled with the flags;
#      run with SR3_LIBRARY=$PWD/tools/bin/libsr3_<name>.so.  Also prints the static loop statistics of kernel $SR3_VARIANT_KERNEL.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/image-super-resolution-via-iterative-refinement_amd/csrc
NAME=$1; SRC=$2; shift 2
mkdir -p $ROOT/tools/bin /tmp/sr3_variant
cd $CS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $SRC.hip -o /tmp/sr3_variant/${SRC}_$NAME.o
OBJS=$(ls build/*.o | grep -v "build/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/sr3_variant/${SRC}_$NAME.o -o $ROOT/tools/bin/libsr3_$NAME.so
echo "built tools/bin/libsr3_$NAME.so ($*)"
if [ -n "$SR3_VARIANT_KERNEL" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" -S --cuda-device-only $SRC.hip -o /tmp/sr3_variant/${SRC}_$NAME.s 2>/dev/null
  python3 $ROOT/tools/isa_loop.py /tmp/sr3_variant/${SRC}_$NAME.s "$SR3_VARIANT_KERNEL" | grep -E "NumVgprs|ScratchSize|innermost|valu_|mfma|lds_|vmem|scratch|detail"
fi
