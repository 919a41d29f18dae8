cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "soil or hot_path or resident or chain" 2>&1 | tail -2
python bench.py --only hotpath --size 5000 --family deep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hotpath 5000:', d['ms_per_model_step'], d['stages_sum_ms'], d['stages']['soil_columns'])"
python bench.py --only soil 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: (v['ms_per_step'], v['frac_hbm'], v['multi_substep_columns_frac']) for k, v in d.items() if isinstance(v, dict) and 'ms_per_step' in v})"
