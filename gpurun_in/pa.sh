cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "pixel_aggregates or hot_path or resident or chain" 2>&1 | tail -2
python bench.py --only hotpath --size 5000 --family deep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hotpath 5000:', d['ms_per_model_step'], d['stages_sum_ms'], {k:(v['ms'],v['frac_hbm']) for k,v in d['stages'].items()})"
