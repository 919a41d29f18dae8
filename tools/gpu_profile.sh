#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of the bench workload.
# Usage: tools/gpu_profile.sh <tag> [bench args...]      outputs -> gpurun_out/prof_<tag>_{kt,fetch,write}
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:-"--size 10000 --steps 5 --warmup 1 --no-extra --no-cpu-baseline --calibrate"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_kt -o kt -- python $ROOT/bench.py $ARGS > $OUT/prof_${TAG}_kt.json 2> $OUT/prof_${TAG}_kt.err; echo rc=$?
echo "== pmc FETCH_SIZE"; timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -o pmc -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/prof_${TAG}_fetch.err; echo rc=$?
echo "== pmc WRITE_SIZE"; timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -o pmc -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/prof_${TAG}_write.err; echo rc=$?
cd $ROOT
python tools/summarize_prof.py $TAG > $OUT/prof_${TAG}_summary.txt 2>&1
cat $OUT/prof_${TAG}_summary.txt
# keep only what is small enough to travel back
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*_agent_info.csv" -delete 2>/dev/null
du -sh $OUT/prof_${TAG}_* | tail -8
