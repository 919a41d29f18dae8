cd $GRAFT_REPO_ROOT
for w in 3 4; do for t in 8 16 24 32 none; do
  lib=""; [ $t != none ] && lib="LISFLOOD_AMD_LIBRARY=$GRAFT_REPO_ROOT/gpurun_in/soil_cap$t.so"
  env $lib LF_SOIL_WAVES=$w LF_BENCH_SOIL_REGIME=wet python bench.py --only soil 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves $w cap $t', d['wet']['ms_per_step'])"
done; done
