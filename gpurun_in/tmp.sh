cd $GRAFT_REPO_ROOT
LF_FUSED_TIME_MAJOR=1 python -m pytest tests -m gpu -x -q -k "fused or wavefront or several_model_steps or sideflow_vector or hot_path" 2>&1 | tail -2
for v in default tm3; do
lib=""; [ $v != default ] && lib="LISFLOOD_AMD_LIBRARY=$GRAFT_REPO_ROOT/gpurun_in/$v.so"
env $lib python bench.py --only hotpath --size 5000 --family deep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v hotpath 5000:', d['ms_per_model_step'], d['stages']['channel_wavefront'])"
env $lib python bench.py --only model_step --size 6000 --family shallow 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v shallow 6000 model step:', d['fused']['ms_per_model_step'])"
done
