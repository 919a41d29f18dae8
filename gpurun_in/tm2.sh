cd $GRAFT_REPO_ROOT
LF_FUSED_TIME_MAJOR=1 python -m pytest tests -m gpu -x -q -k "fused or wavefront or hot_path or resident or model_step or substep or compact" 2>&1 | tail -2
for tm in 1; do
LF_FUSED_TIME_MAJOR=$tm python bench.py --only hotpath --size 5000 --family deep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tm=$tm hotpath 5000:', d['ms_per_model_step'], d['one_stream_ms_per_model_step'], d['stages']['channel_wavefront'])"
LF_FUSED_TIME_MAJOR=$tm python bench.py --only model_step --size 6000 --family shallow 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tm=$tm shallow 6000 model step:', {k:(v['ms_per_model_step'],v['launches_per_model_step']) for k,v in d.items() if isinstance(v,dict) and 'ms_per_model_step' in v})"
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pq; LF_FUSED_TIME_MAJOR=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o p -- python $GRAFT_REPO_ROOT/bench.py --only model_step --size 6000 --family shallow > /dev/null 2>&1; f=$(find /tmp/pq -name "*kernel_stats.csv" | head -1); head -4 $f | cut -c1-150
