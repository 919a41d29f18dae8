#!/bin/bash
# Arbitrary counter passes for one workload of tools/pmc_route.py (runs on the GPU box via gpurun), one rocprofv3 --pmc run per
# pass, per-kernel totals -> gpurun_out/pmc_<tag>_<workload>.txt
#   tools/pmc_passes.sh tag mode:family:size:reps "COUNTER COUNTER ..." ["COUNTER ..." ...]
set -u
TAG=$1; W=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IFS=: read mode fam size reps <<< "$W"
name=${mode}_${fam}_${size}
i=0
for C in "$@"; do
  i=$((i+1)); rm -rf /tmp/pp_${TAG}_$i
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pp_${TAG}_$i -o pmc -- \
      python $ROOT/tools/pmc_route.py $mode $fam $size $reps > /tmp/pp_${TAG}_$i.log 2>&1
  echo "pass $i rc=$? $(grep -i "error\|invalid\|not found" /tmp/pp_${TAG}_$i.log | head -2)"
done
python - "$TAG" "$name" > $OUT/pmc_${TAG}_${name}.txt <<'PY'
import csv, glob, collections, re, sys
tag, name = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/pp_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"\(.*", "", k)
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
print("# %s %s: per-kernel counter totals over the run (rocprofv3 --pmc, one pass per counter group)" % (tag, name))
for k, c in sorted(agg.items()):
    n = max(v[0] for v in c.values())
    print("  ".join(["%-44s launches=%d" % (k, n)] + ["%s=%.6g" % (cn, v[1]) for cn, v in sorted(c.items())]))
PY
grep -v "rocclr\|k_check" $OUT/pmc_${TAG}_${name}.txt
