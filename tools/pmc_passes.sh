#!/bin/bash
# The three counter passes of a round (tools/evidence_pass1.sh, tools/round_profile.sh) on their own (FETCH_SIZE, WRITE_SIZE, SQ set; each its own run, --kernel-trace only):
#   bash tools/pmc_passes.sh <tag> [extra bench.py flags, e.g. --exact-fp32]      (through gpurun; writes gpurun_out/<tag>/{fetch,write,sq} + the summaries of pmc_traffic.py / pmc_sq.py)
set -u
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
shift || true
Q="--no-cpu-baseline --no-torch-baseline --train-steps 0 --no-split-leg --no-other-configs --no-exact-leg --no-roofline $*"
for attempt in 1 2; do
  rm -rf $OUT/fetch
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- python bench.py --steps 4 --warmup 1 $Q > /dev/null 2> $OUT/fetch.err && break
  echo "fetch pass attempt $attempt failed"; tail -3 $OUT/fetch.err | cut -c1-200
done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- python bench.py --steps 4 --warmup 1 $Q > /dev/null 2> $OUT/write.err; echo "write rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o bench -- python bench.py --steps 4 --warmup 1 $Q > /dev/null 2> $OUT/sq.err; echo "sq rc=$?"
python tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/$TAG > $OUT/pmc.log 2>&1
python tools/pmc_sq.py $OUT/sq $OUT/$TAG >> $OUT/pmc.log 2>&1
tail -5 $OUT/pmc.log
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete
ls $OUT
