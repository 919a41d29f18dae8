cd $GRAFT_REPO_ROOT
for v in default tm0; do
lib=""; [ $v != default ] && lib="LISFLOOD_AMD_LIBRARY=$GRAFT_REPO_ROOT/gpurun_in/$v.so"
env $lib LF_FUSED_TIME_MAJOR=1 python bench.py --only hotpath --size 5000 --family deep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v hotpath 5000:', d['ms_per_model_step'], d['stages']['channel_wavefront'])"
env $lib LF_FUSED_TIME_MAJOR=1 python bench.py --only model_step --size 6000 --family shallow 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v shallow 6000 model step:', d['fused']['ms_per_model_step'])"
done
