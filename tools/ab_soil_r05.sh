#!/bin/bash
# Round-5 A/B of the soil kernel on the GPU box.  usage: ab_soil_r05.sh "<ENV=VAL ...>" "<ENV=VAL ...>" ...
# each argument is one variant (environment of the run, e.g. "LF_SOIL_TILE=64" or "LF_SOIL_TWO_PASS=1"): the soil parity
# tests under that environment, then `bench.py --only soil` twice.  Same box, back to back.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
[ $# -eq 0 ] && set -- ""
for v in "$@"; do
  tag=$(echo "$v" | tr ' =' '__'); [ -z "$tag" ] && tag=default
  echo "== variant: ${v:-default}"
  env $v python -m pytest tests -m gpu -x -q -k "soil_columns or soilloop or soil_full" 2>&1 | tail -2
  for rep in 1 2; do
    env $v python bench.py --only soil > $OUT/r05_soil_${tag}_$rep.json 2> $OUT/r05_soil_${tag}_$rep.err
    python - $OUT/r05_soil_${tag}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   ", {k: (v["ms_per_step"], v["frac_hbm"], v["multi_substep_columns_frac"]) for k, v in d.items() if isinstance(v, dict) and "ms_per_step" in v})
PY
  done
done
