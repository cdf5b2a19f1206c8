#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/j22; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_ops.py -x -q -k "stale_derived or without_clipping or p_sample or test_conv" 2>&1 | tail -4 | tee $O/pytest_sub.txt
timeout 600 python -m pytest tests/test_gpu_trajectory.py -x -q -s -k "c2" 2>&1 | grep -E "steps:|CPU oracle|passed|failed" | tee $O/trajectory_c2.txt
timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-other-configs > $O/bench_quick.json 2> $O/bench_quick.err; python - <<'PY'
import json
b=json.load(open('gpurun_out/j22/bench_quick.json'))
r=b['roofline']
print(b['ms_per_step'], r['frac'], r['traffic'], r['algorithmic_bytes_per_launch'], r['traffic_over_algorithmic'], r['sq_counters'] and {k:(v and round(v.get('mfma_busy',0),3)) for k,v in r['sq_counters'].items()})
PY
