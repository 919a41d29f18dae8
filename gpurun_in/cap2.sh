cd $GRAFT_REPO_ROOT
for cap in 8 16 24 40; do
LF_SOIL_TRIP_CAP=$cap python bench.py --only hotpath --size 4000 --family deep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cap=$cap hotpath 4000:', d['ms_per_model_step'], d['stages']['soil_columns'])"
done
