#!/bin/bash
# round-4 gradient-noise investigation, second pass: which approximation is it (v_rcp_f32 / v_exp_f32), and the gamma extremes
set -u
OUT=gpurun_out/r04b
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=$PWD/image-super-resolution-via-iterative-refinement_amd/csrc
P="python tools/grad_probe.py"
timeout 300 $P --batch 64 --gamma uniform --data-seed 8 --ref-cache /tmp/ref_u8.pt --top 4 --variant default --variant dgrad_winograd=0 --variant winograd=0 > $OUT/uniform8_nr.txt 2>&1; echo "nr rc=$?"
for n in fastrcp winoexpf exact; do
  SR3_LIBRARY=$C/build_$n/libsr3_$n.so timeout 300 $P --batch 64 --gamma uniform --data-seed 8 --ref-cache /tmp/ref_u8.pt --top 4 --variant default > $OUT/uniform8_$n.txt 2>&1; echo "$n rc=$?"
done
timeout 300 $P --batch 32 --gamma high --data-seed 9 --f32 --top 4 --ref-cache /tmp/ref_h9.pt --variant default --variant dgrad_winograd=0 --variant winograd=0 > $OUT/high9_nr.txt 2>&1; echo "high rc=$?"
SR3_LIBRARY=$C/build_exact/libsr3_exact.so timeout 300 $P --batch 32 --gamma high --data-seed 9 --top 4 --ref-cache /tmp/ref_h9.pt --variant default --variant winograd=0 > $OUT/high9_exact.txt 2>&1; echo "high exact rc=$?"
timeout 300 $P --batch 32 --gamma low --data-seed 10 --f32 --top 4 --variant default --variant winograd=0 > $OUT/low10_nr.txt 2>&1; echo "low rc=$?"
timeout 300 $P --batch 64 --gamma mixed --data-seed 11 --f32 --top 4 --variant default --variant winograd=0 > $OUT/mixed11_nr.txt 2>&1; echo "mixed rc=$?"
grep -h "^engine\|^oracle" $OUT/*.txt | cut -c1-200
