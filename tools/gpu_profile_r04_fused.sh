#!/bin/bash
# Round-4 evidence for the fused model step (runs on the GPU box via gpurun): rocprofv3 --kernel-trace --stats of
# tools/pmc_route.py fused deep <size> 2 (24 split sub-steps per model step) with the per-launch choice (default) and with
# the one-wavefront cone kernel forced (LF_FUSED_SPLIT=0) -> gpurun_out/r04_fused_deep_<size>_{auto,onewave}_kernel_stats.csv
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for size in ${@:-2000}; do
  for mode in auto onewave; do
    rm -rf /tmp/kt_$mode
    if [ $mode = onewave ]; then export LF_FUSED_SPLIT=0; else unset LF_FUSED_SPLIT; fi
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$mode -o kt -- \
        python $ROOT/tools/pmc_route.py fused deep $size 2 > /tmp/kt_$mode.log 2>&1
    echo "$size $mode rc=$? $(tail -1 /tmp/kt_$mode.log)"
    f=$(find /tmp/kt_$mode -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $OUT/r04_fused_deep_${size}_${mode}_kernel_stats.csv
    head -6 $OUT/r04_fused_deep_${size}_${mode}_kernel_stats.csv | cut -c1-200
  done
done
