cd $GRAFT_REPO_ROOT
for nr in 0 1; do
LF_NO_RECOMPUTE=$nr python bench.py --only model_step 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_recompute=$nr deep 5000', {k:(v.get('ms_per_model_step')) for k,v in d.items() if isinstance(v,dict) and k.startswith('fused') and k!='fused_level_by_level'})"
LF_NO_RECOMPUTE=$nr python bench.py --only structures 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_recompute=$nr structures 3000', d['fused']['ms_per_model_step'])"
done
