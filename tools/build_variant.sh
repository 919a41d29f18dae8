#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." [source stem, default lf_router]: a copy of the library with one translation unit
# rebuilt under extra flags -> gpurun_in/NAME.so (same-box A/B through LISFLOOD_AMD_LIBRARY; gpurun_in/ is not tracked)
set -e
cd "$(dirname "$0")/../lisflood-code_amd"
n=$1; f=$2; stem=${3:-lf_router}
mkdir -p /tmp/var ../gpurun_in
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 $f -x hip -c csrc/$stem.hip -o /tmp/var/${stem}_$n.o 2>/dev/null
objs=""
for o in lf_device.cpp lf_graph.cpp lf_router.hip lf_soil.hip lf_dist.hip lf_modules.hip lf_ldd.hip; do
  if [ "$o" = "$stem.hip" ]; then objs="$objs /tmp/var/${stem}_$n.o"; else objs="$objs build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../gpurun_in/$n.so $objs -ldl
echo built $n
