cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "hot_path or resident or warm" 2>&1 | tail -2
python bench.py --only hotpath --size 5000 --family river 2>gpurun_out/hp4.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v for k,v in d.items() if k.startswith('ms_') or k.startswith('one_') or k.startswith('forcing') or k=='stages_sum_ms'})"
tail -2 gpurun_out/hp4.err
