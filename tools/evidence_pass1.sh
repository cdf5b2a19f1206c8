#!/bin/bash
# Round evidence, pass 1 of 2 (through gpurun): everything that must exist under profiles/ BEFORE the final bench line is taken,
# because bench.py reads the counter summaries of the round from there -- smoke, rocprofv3 kernel statistics of the sampling leg,
# the counter passes (FETCH_SIZE, WRITE_SIZE, SQ set; default and exact-fp32 plans), the per-launch table, kernel statistics of a
# training step and of stock PyTorch-ROCm.  Pass 2 = tools/evidence_pass2.sh <tag> (bench line, 1-rank launcher lines, the -m gpu suite).
#   bash tools/evidence_pass1.sh r06
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
Q="--no-cpu-baseline --no-torch-baseline --train-steps 0 --no-split-leg --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --steps 200 --warmup 5 $Q > $OUT/bench_under_rocprof.json 2> $OUT/stats.err; echo "stats rc=$?"
bash tools/pmc_passes.sh $TAG
bash tools/pmc_passes.sh ${TAG}_exact_fp32 --exact-fp32
python tools/op_table.py > $OUT/op_table.txt 2> $OUT/op_table.err; tail -14 $OUT/op_table.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -o train -- python tools/gpu_probe.py --train 64 > $OUT/train_probe.log 2>&1; grep -h "train_step" $OUT/train_probe.log | cut -c1-200
# SQ counters of the SAME training step (own pass, --kernel-trace only): bench.py's train.roofline reads <tag>_train_sq_counters.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/train_sq -o train -- python tools/gpu_probe.py --train 64 > $OUT/train_sq.log 2>&1; echo "train sq rc=$?"
python tools/pmc_sq.py $OUT/train_sq $OUT/${TAG}_train >> $OUT/pmc.log 2>&1
python tools/torch_baseline_probe.py --config sr3_16_128 --batch 16 --steps 3 > $OUT/torch_warm.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/torch_stats -o torch -- python tools/torch_baseline_probe.py --config sr3_16_128 --batch 16 --steps 5 > $OUT/torch_probe.log 2>&1
# afterwards, here: copy the summaries into profiles/ (see profiles/README.md) and run  python tools/merge_counters.py profiles/$TAG
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete
ls $OUT | head -40
