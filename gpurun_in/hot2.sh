cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "hot_path or resident" 2>&1 | tail -2
python bench.py --only hotpath --size 5000 --family deep 2> gpurun_out/r05_hot2.err > gpurun_out/r05_hot2.json; tail -1 gpurun_out/r05_hot2.err; python -c "
import json; d=json.load(open('gpurun_out/r05_hot2.json')); print(d['ms_per_model_step'], d['one_stream_ms_per_model_step'], d['stages_sum_ms'])"
