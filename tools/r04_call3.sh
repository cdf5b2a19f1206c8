#!/bin/bash
# round-4 call 3: new parity tests (float64 gradients, unconditional SR3, 8x8 Winograd tile), de-kinked gradient probe, quick bench
set -u
OUT=gpurun_out/r04c
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "winograd or wino or dropout or fused_output_stats" > $OUT/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $OUT/pytest_ops.log
timeout 1500 python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_unet.py tests/test_gpu_train.py -q -s -k "bench_configs or uncond" > $OUT/pytest_cfg.log 2>&1; echo "cfg rc=$?"; tail -3 $OUT/pytest_cfg.log
P="python tools/grad_probe.py --kink-margin 1e-4 --top 3"
timeout 300 $P --batch 64 --gamma uniform --data-seed 8 --f32 --variant default --variant winograd=0 > $OUT/probe_uniform8.txt 2>&1; echo "probe rc=$?"
timeout 300 $P --batch 32 --gamma high --data-seed 9 --f32 --variant default --variant winograd=0 > $OUT/probe_high9.txt 2>&1
timeout 300 $P --batch 32 --gamma low --data-seed 10 --f32 --variant default --variant winograd=0 > $OUT/probe_low10.txt 2>&1
grep -h "^engine\|^oracle\|de-kinked" $OUT/probe_*.txt | cut -c1-170
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-torch-baseline --train-steps 3 --no-other-configs > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04c/bench_quick.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'train', d.get('train',{}).get('ms_per_step'), 'parity', d.get('parity'))
PY
timeout 300 python tools/op_table.py > $OUT/op_table.txt 2> $OUT/op_table.err; tail -25 $OUT/op_table.txt
