cd $GRAFT_REPO_ROOT
LF_FUSED_TIME_MAJOR=1 python -m pytest tests -m gpu -x -q -k "fused or wavefront or several_model_steps or sideflow_vector or hot_path or dist or row_block" 2>&1 | tail -2
python bench.py --only hotpath --size 5000 --family deep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hotpath 5000:', d['ms_per_model_step'], d['stages']['channel_wavefront'])"
python bench.py --only model_step --size 6000 --family shallow 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shallow 6000 model step:', d['fused']['ms_per_model_step'])"
