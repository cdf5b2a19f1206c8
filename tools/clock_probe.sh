#!/bin/bash
# Sample the shader clock and power while the sampling leg runs (is the fp32 MFMA peak of 2.4 GHz x 256 CUs actually
# available under this kernel mix?).  Usage (through gpurun): bash tools/clock_probe.sh
OUT=gpurun_out/clock_probe.txt
mkdir -p gpurun_out
: > $OUT
( for i in $(seq 1 40); do
    echo "t=$i" >> $OUT
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" >> $OUT
    sleep 0.5
  done ) &
MON=$!
sleep 2
python bench.py --steps 800 --warmup 10 --no-cpu-baseline --train-steps 0 --no-split-leg --no-roofline > gpurun_out/clock_probe_bench.json 2>/dev/null
wait $MON
grep -E "sclk|Power|power" $OUT | sort | uniq -c | sort -rn | head -20
cat gpurun_out/clock_probe_bench.json | cut -c1-200
