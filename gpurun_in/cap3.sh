cd $GRAFT_REPO_ROOT
for cap in 4 6 8 10 12; do
LF_SOIL_TRIP_CAP=$cap python bench.py --only hotpath --size 4000 --family deep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cap=$cap hotpath 4000 soil:', d['stages']['soil_columns']['ms'])"
LF_SOIL_TRIP_CAP=$cap LF_BENCH_SOIL_REGIME=wet python bench.py --only soil 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cap=$cap bench wet:', d['wet']['ms_per_step'])"
done
