cd $GRAFT_REPO_ROOT
for v in default cones3 cones4; do
lib=""; [ $v != default ] && lib="LISFLOOD_AMD_LIBRARY=$GRAFT_REPO_ROOT/gpurun_in/$v.so"
env $lib python bench.py --only model_step 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', {k:(v.get('ms_per_model_step')) for k,v in d.items() if isinstance(v,dict) and k.startswith('fused') and k!='fused_level_by_level'})"
done
