cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "several_model_steps or fused or wavefront or model_step" 2>&1 | tail -3
python bench.py --only model_step 2>gpurun_out/r05_ms.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(v.get('ms_per_model_step'),v.get('launches_per_model_step'), v.get('error')) for k,v in d.items() if isinstance(v,dict)})"
tail -2 gpurun_out/r05_ms.err
