#!/bin/bash
# Round-2 profile set (runs on the GPU box via gpurun):
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command           -> prof_<tag>_kt
#   2. FETCH_SIZE / WRITE_SIZE passes (separate runs) of the routing workload  -> prof_<tag>_{fetch,write}
#   3. the same three passes of the soil kernel (wet regime, staged pass 2)    -> prof_<tag>soil_{kt,fetch,write}
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace of the default bench command"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_kt -o kt -- python $ROOT/bench.py > $OUT/prof_${TAG}_kt.json 2> $OUT/prof_${TAG}_kt.err; echo rc=$?
echo "== kernel trace of the headline workload alone (k_level mean for roofline.frac)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}head_kt -o kt -- python $ROOT/bench.py --no-extra --no-cpu-baseline > $OUT/prof_${TAG}head_kt.json 2> $OUT/prof_${TAG}head_kt.err; echo rc=$?
R="--size 10000 --steps 5 --warmup 1 --no-extra --no-cpu-baseline --calibrate"
echo "== pmc FETCH_SIZE (routing)"; timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -o pmc -- python $ROOT/bench.py $R > /dev/null 2> $OUT/prof_${TAG}_fetch.err; echo rc=$?
echo "== pmc WRITE_SIZE (routing)"; timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -o pmc -- python $ROOT/bench.py $R > /dev/null 2> $OUT/prof_${TAG}_write.err; echo rc=$?
S="--only soil --steps 5 --warmup 3 --no-cpu-baseline"
echo "== soil kernel trace"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}soil_kt -o kt -- python $ROOT/bench.py $S > $OUT/prof_${TAG}soil_kt.json 2> $OUT/prof_${TAG}soil_kt.err; echo rc=$?
echo "== soil FETCH_SIZE"; timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}soil_fetch -o pmc -- python $ROOT/bench.py $S > /dev/null 2> $OUT/prof_${TAG}soil_fetch.err; echo rc=$?
echo "== soil WRITE_SIZE"; timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}soil_write -o pmc -- python $ROOT/bench.py $S > /dev/null 2> $OUT/prof_${TAG}soil_write.err; echo rc=$?
cd $ROOT
python tools/summarize_prof.py $TAG > $OUT/prof_${TAG}_summary.txt 2>&1
python tools/summarize_prof.py ${TAG}soil > $OUT/prof_${TAG}soil_summary.txt 2>&1
python tools/summarize_prof.py ${TAG}head > $OUT/prof_${TAG}head_summary.txt 2>&1
head -12 $OUT/prof_${TAG}head_summary.txt
tail -60 $OUT/prof_${TAG}soil_summary.txt
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*_agent_info.csv" -delete 2>/dev/null
find $OUT -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
find $OUT -name "*counter_collection.csv" -size +8M -delete 2>/dev/null
du -sh $OUT/prof_${TAG}* | tail -12
