cd $GRAFT_REPO_ROOT
LF_FUSED_LEAN=1 python -m pytest tests -m gpu -x -q -k "fused or wavefront or several_model_steps or model_step or hot_path or config4_workload_8000" 2>&1 | tail -2
for lean in 0 auto 1; do
e=""; [ $lean != auto ] && e="LF_FUSED_LEAN=$lean"
env $e python bench.py --only model_step 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lean=$lean', {k:(v.get('ms_per_model_step')) for k,v in d.items() if isinstance(v,dict) and k.startswith('fused') and k!='fused_level_by_level'})"
done
