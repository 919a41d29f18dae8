cd $GRAFT_REPO_ROOT
for v in default soil_nt default soil_nt; do
lib=""; [ $v != default ] && lib="LISFLOOD_AMD_LIBRARY=$GRAFT_REPO_ROOT/gpurun_in/$v.so"
env $lib python bench.py --only soil 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', {k: (v['ms_per_step'], v['frac_hbm']) for k, v in d.items() if isinstance(v, dict) and 'ms_per_step' in v})"
done
