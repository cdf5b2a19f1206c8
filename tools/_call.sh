cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "attention" -s 2>&1 | grep -E "attention B|passed|failed|Error|assert" | head -30
timeout 120 python tools/op_table.py 2>/dev/null | tail -12
timeout 120 python tools/op_table.py --opt attn_split=0 2>/dev/null | grep "#   60"
