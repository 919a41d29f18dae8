cd /tmp && export TMPDIR=/tmp
for regime in wet single_substep; do
rm -rf /tmp/ps_$regime
LF_BENCH_SOIL_REGIME=$regime rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$regime -o p -- python $GRAFT_REPO_ROOT/bench.py --only soil > /tmp/ps_$regime.log 2>&1
f=$(find /tmp/ps_$regime -name "*kernel_stats.csv" | head -1)
echo "== $regime"; head -6 $f | cut -c1-200
cp $f $GRAFT_REPO_ROOT/gpurun_out/r05b_soil_${regime}_kernel_stats.csv
done
