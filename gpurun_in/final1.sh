cd $GRAFT_REPO_ROOT
time python bench.py --steps 20 --warmup 5 > gpurun_out/r05b_bench.json 2> gpurun_out/r05b_bench.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/r05b_bench_detail.json
wc -c gpurun_out/r05b_bench.json; cat gpurun_out/r05b_bench.json
python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r05b_pytest_gpu_full.log
