#!/bin/bash
# Round-4 counter evidence for the cone sweep (runs on the GPU box via gpurun): the round-3 recipe (tools/pmc_r03.sh:
# three separate rocprofv3 --pmc passes per workload) for the supply/chain kernel (default) and, beside it, for the
# one-wavefront-per-cone kernel of round 3 (LF_ROUTE_SPLIT=0) on the same box -> gpurun_out/pmc_r04a_* and pmc_r04a1w_*
set -u
cd "$(dirname "$0")/.."
WL=${@:-"route:deep:10000:2 route:river:10000:2"}
tools/pmc_r03.sh r04a $WL
LF_ROUTE_SPLIT=0 tools/pmc_r03.sh r04a1w $WL
