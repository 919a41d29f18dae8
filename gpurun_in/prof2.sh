cd /tmp && export TMPDIR=/tmp
for what in overland hotpath; do
rm -rf /tmp/pp_$what
extra=""; [ $what = hotpath ] && extra="--size 5000 --family deep"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$what -o p -- python $GRAFT_REPO_ROOT/bench.py --only $what $extra > /tmp/pp_$what.log 2>&1
f=$(find /tmp/pp_$what -name "*kernel_stats.csv" | head -1)
echo "== $what"; python - $f <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n)
    print("%-60s calls %6s total %9.3f ms avg %9.1f us  %5s%%" % (n[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cp $f $GRAFT_REPO_ROOT/gpurun_out/r05_${what}_kernel_stats.csv
done
