#!/bin/bash
# Round-6 profile evidence (runs on the GPU box via gpurun; outputs under gpurun_out/, summaries copied to profiles/ by hand):
#   1. rocprofv3 --kernel-trace --stats of the default bench command            -> gpurun_out/prof_r06head_kt/
#   2. FETCH_SIZE / WRITE_SIZE passes (one rocprofv3 run per counter, as MI355X_MICROARCH.md prescribes) over the
#      land-surface stages of the resident step: fused / separate / lean       -> gpurun_out/pmc_r06_land_<mode>.txt
#   3. the same two passes over the model step with structures                  -> gpurun_out/pmc_r06_structures.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
SIZE=${1:-3000}
cd /tmp && export TMPDIR=/tmp
agg() { # agg <glob dir prefix> <label>
python - "$1" "$2" <<'PY'
import csv, glob, collections, re, sys
prefix, label = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(prefix + "*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"\(.*", "", k).replace("void ", "")
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
print("# %s: per-kernel counter totals over the run (rocprofv3 --pmc, one pass per counter); HBM bytes = FETCH_SIZE [KiB] x 2"
      " (gfx950: the counter reads half, calibrated with a stream copy in round 1) + WRITE_SIZE [KiB]" % label)
for k, c in sorted(agg.items()):
    n = max(v[0] for v in c.values())
    rd = c.get("FETCH_SIZE", [0, 0.0])[1] * 1024 * 2      # KiB; x2: the gfx950 calibration of the read counter
    wr = c.get("WRITE_SIZE", [0, 0.0])[1] * 1024
    print("%-60s launches=%-6d read_GB=%-10.4f write_GB=%-10.4f total_GB=%.4f" % (k, n, rd / 1e9, wr / 1e9, (rd + wr) / 1e9))
PY
}
echo "== 0. kernel trace of the headline alone (the default size, 100 calls, no other leg: the k_level launches of this run are the headline's)"
rm -rf $OUT/prof_r06head_only
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r06head_only -o kt -- python $ROOT/bench.py --no-extra --no-cpu-baseline > $OUT/prof_r06head_only_bench.json 2> /dev/null; echo rc=$?
find $OUT/prof_r06head_only -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
echo "== 1. kernel trace of the default bench command"
rm -rf $OUT/prof_r06head_kt
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r06head_kt -o kt -- python $ROOT/bench.py --no-cpu-baseline > $OUT/prof_r06head_bench.json 2> $OUT/prof_r06head_bench.err; echo rc=$?
for mode in fused separate lean; do
  echo "== 2. land surface, $mode"
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pl_${mode}_$C
    timeout 900 rocprofv3 --pmc $C --output-format csv -d /tmp/pl_${mode}_$C -o pmc -- python $ROOT/tools/pmc_land.py $mode $SIZE 2 > /tmp/pl_${mode}_$C.log 2>&1; echo "  $C rc=$?"
  done
  tail -1 /tmp/pl_${mode}_FETCH_SIZE.log > $OUT/pmc_r06_land_${mode}.txt
  agg /tmp/pl_${mode}_ "land surface $mode ${SIZE}^2, 4 model steps" | grep -v "rocclr\|k_check\|k_gather\|k_inert" >> $OUT/pmc_r06_land_${mode}.txt
done
echo "== 3. model step with structures"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ps_$C
  timeout 900 rocprofv3 --pmc $C --output-format csv -d /tmp/ps_$C -o pmc -- python $ROOT/bench.py --only structures > /tmp/ps_$C.log 2>&1; echo "  $C rc=$?"
done
tail -c 600 /tmp/ps_FETCH_SIZE.log > $OUT/pmc_r06_structures.txt; echo >> $OUT/pmc_r06_structures.txt
agg /tmp/ps_ "bench.py --only structures (3000^2 deep, 64 lakes + 192 reservoirs)" | grep -v rocclr >> $OUT/pmc_r06_structures.txt
find $OUT/prof_r06head_kt -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null   # (the raw trace is ~70 MB: over the merge cap)
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*_agent_info.csv" -delete 2>/dev/null
du -sh $OUT/prof_r06head_kt $OUT/pmc_r06_* | tail
