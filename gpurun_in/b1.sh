cd $GRAFT_REPO_ROOT
python -m pytest tests/test_real_rccl_gpu.py -m gpu -x -q 2>&1 | tail -3
MASTER_ADDR=127.0.0.1 python bench.py --force-dist --steps 10 --warmup 2 > gpurun_out/r05_force_dist_1rank.json 2> gpurun_out/r05_force_dist_1rank.err; echo "force-dist rc=$?"; cut -c1-600 gpurun_out/r05_force_dist_1rank.json
time python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/r05_bench_detail.json
wc -c gpurun_out/r05_bench.json; cat gpurun_out/r05_bench.json
