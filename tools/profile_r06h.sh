set -u
cd $GRAFT_REPO_ROOT
bash tools/pmc_r03.sh r06h "route:shallow:10000:4" 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/prof_r06h_only
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r06h_only -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline > $OUT/prof_r06h_only_bench.json 2> /dev/null; echo rc=$?
find $OUT/prof_r06h_only -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
find $OUT/prof_r06h_only -name "*kernel_stats.csv" | head; head -5 $(find $OUT/prof_r06h_only -name "*kernel_stats.csv" | head -1)
