#!/bin/bash
# After tools/evidence_pass1.sh <tag> came back (gpurun_out/<tag>, gpurun_out/<tag>_exact_fp32): copy the summaries bench.py and the docs read into
# profiles/ and merge the exact-fp32 plan's kernels into the counter tables.  Then: tools/evidence_pass2.sh <tag> through gpurun.
#   bash tools/copy_pass1.sh r06
set -e
TAG=${1:-r06}
G=gpurun_out
cp $G/$TAG/stats/bench_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
cp $G/$TAG/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
for f in bench_hbm_pmc.csv hbm_traffic.json sq_counters.csv sq_counters.json train_sq_counters.csv train_sq_counters.json; do cp $G/$TAG/${TAG}_$f profiles/${TAG}_$f; done
for f in bench_hbm_pmc.csv hbm_traffic.json sq_counters.csv sq_counters.json; do cp $G/${TAG}_exact_fp32/${TAG}_exact_fp32_$f profiles/${TAG}_exact_fp32_$f; done
cp $G/$TAG/op_table.txt profiles/${TAG}_op_table_c2_b16.txt
cp $G/$TAG/train_stats/train_kernel_stats.csv profiles/${TAG}_train_kernel_stats.csv
grep "train_step" $G/$TAG/train_probe.log > profiles/${TAG}_train_probe.txt
cp $G/$TAG/torch_stats/torch_kernel_stats.csv profiles/${TAG}_torch_rocm_kernel_stats.csv
python tools/merge_counters.py profiles/$TAG
