#!/bin/bash
# Round-3 kernel-trace set (runs on the GPU box via gpurun):
#   1. rocprofv3 --kernel-trace --stats of the headline workload alone  -> gpurun_out/prof_<tag>head_kt  (k_level mean)
#   2. the same of the DEFAULT bench command                            -> gpurun_out/prof_<tag>_kt
# Counter passes: tools/pmc_r03.sh (separate runs, never together with a trace).
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace of the headline workload alone"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}head_kt -o kt -- python $ROOT/bench.py --no-extra --no-cpu-baseline > $OUT/prof_${TAG}head_kt.json 2> $OUT/prof_${TAG}head_kt.err; echo rc=$?
echo "== kernel trace of the default bench command"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_kt -o kt -- python $ROOT/bench.py > $OUT/prof_${TAG}_kt.json 2> $OUT/prof_${TAG}_kt.err; echo rc=$?
cd $ROOT
python tools/summarize_prof.py ${TAG}head > $OUT/prof_${TAG}head_summary.txt 2>&1
python tools/summarize_prof.py ${TAG} > $OUT/prof_${TAG}_summary.txt 2>&1
head -14 $OUT/prof_${TAG}head_summary.txt
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*_agent_info.csv" -delete 2>/dev/null
find $OUT -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
for t in ${TAG}head ${TAG}; do f=$(find $OUT/prof_${t}_kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${t}_kernel_stats.csv; done
ls -la $OUT/${TAG}*_kernel_stats.csv
