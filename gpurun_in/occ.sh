cd $GRAFT_REPO_ROOT
for lds in 0 20000 46000; do for cap in 16 0; do
  env LF_SOIL_DEBUG_LDS=$lds LF_SOIL_TRIP_CAP=$cap python bench.py --only soil 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('extra lds $lds cap $cap', d['wet']['ms_per_step'], d['single_substep']['ms_per_step'])"
done; done
