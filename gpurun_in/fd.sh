cd $GRAFT_REPO_ROOT
python -m pytest tests/test_real_rccl_gpu.py tests/test_dist_multirank_gpu.py -m gpu -x -q 2>&1 | tail -2
MASTER_ADDR=127.0.0.1 python bench.py --force-dist --steps 10 --warmup 2 > gpurun_out/r05_force_dist_1rank.json 2> gpurun_out/r05_force_dist_1rank.err; echo "force-dist rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/r05_force_dist_1rank.json').read().strip().splitlines()[-1]); m=d['model_step_24_substeps_split_row_blocks']; print(d['value'], {k:m[k] for k in ('ms_per_model_step','max_launches_per_model_step','several_model_steps_per_call')}, d['row_block_vs_catchment_partition_sumQ_rel_diff'], d['rccl_library'])"
