#!/bin/bash
# Round-6 final kernel traces (GPU box, via gpurun): rocprofv3 --kernel-trace --stats of the default bench command
# -> gpurun_out/prof_r06f_kt/kt_kernel_stats.csv (copied to profiles/r06_kernel_stats.csv) and its bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_r06f_kt
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r06f_kt -o kt -- python $ROOT/bench.py --no-cpu-baseline > $OUT/prof_r06f_bench.json 2> $OUT/prof_r06f_bench.err; echo rc=$?
find $OUT/prof_r06f_kt -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
find $OUT -name "*.db" -delete 2>/dev/null
head -12 $(find $OUT/prof_r06f_kt -name "*kernel_stats.csv" | head -1)
