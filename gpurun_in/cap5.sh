cd $GRAFT_REPO_ROOT
for cap in 6 16; do
LF_SOIL_TRIP_CAP=$cap LF_BENCH_SOIL_REGIME=wet python -c "
import bench, json
d = bench.soil_bench(N=16_000_000, steps=6)
print('cap=$cap N=16M px:', d['wet']['ms_per_step'], d['wet']['multi_substep_columns_frac'])"
done
