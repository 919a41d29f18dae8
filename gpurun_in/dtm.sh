cd $GRAFT_REPO_ROOT
for tm in 1 0; do
echo "== LF_FUSED_TIME_MAJOR=$tm"
LF_FUSED_TIME_MAJOR=$tm python -m pytest tests/test_dist_fused_gpu.py tests/test_dist_multirank_gpu.py -m gpu -x -q 2>&1 | tail -2
done
python tools/bench_dist_model_steps.py 5000 4 2>&1 | tail -2
