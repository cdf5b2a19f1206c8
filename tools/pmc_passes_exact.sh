#!/bin/bash
set -u
OUT=gpurun_out/r04x; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 560 python -m pytest tests/test_gpu_bench_configs.py -q -x -m gpu -s -k "c3_train or c5_train or c4_batch4 or c5_batch32" > $OUT/pytest_configs_rest.log 2>&1; echo "configs rc=$?"; tail -1 $OUT/pytest_configs_rest.log
Q="--no-cpu-baseline --no-torch-baseline --train-steps 0 --no-split-leg --no-other-configs --no-exact-leg --no-roofline --exact-fp32"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- python bench.py --steps 4 --warmup 1 $Q > /dev/null 2> $OUT/fetch.err; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- python bench.py --steps 4 --warmup 1 $Q > /dev/null 2> $OUT/write.err; echo "write rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o bench -- python bench.py --steps 4 --warmup 1 $Q > /dev/null 2> $OUT/sq.err; echo "sq rc=$?"
python tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/r04x > $OUT/pmc.log 2>&1
python tools/pmc_sq.py $OUT/sq $OUT/r04x >> $OUT/pmc.log 2>&1
tail -3 $OUT/pmc.log
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete
