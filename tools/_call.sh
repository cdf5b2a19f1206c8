cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--steps 300 --warmup 10 --no-cpu-baseline --no-torch-baseline --train-steps 0 --no-other-configs --no-exact-leg --no-roofline"
for o in 1 0 1 0; do
timeout 200 python bench.py $Q --plan-opt conv_fold=$o 2>/dev/null | python -c "import sys,json; print('conv_fold=$o', json.loads(sys.stdin.read())['ms_per_step'])"
done
