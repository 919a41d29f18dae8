#!/bin/bash
# Round-5 A/B of the soil kernels on the GPU box: the one-launch form (k_soil_fused, default) against the two-pass form
# (LF_SOIL_TWO_PASS=1), same box, same process layout: parity tests first, then `bench.py --only soil` of each.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q -k "soil or canopy or chain or hot_path or structures_mid_size or cone or deep" 2>&1 | tail -5 | tee $OUT/r05_soil_tests.txt
for tp in 0 1; do
  for rep in 1 2; do
    LF_SOIL_TWO_PASS=$tp python bench.py --only soil > $OUT/r05_soil_bench_tp${tp}_$rep.json 2> $OUT/r05_soil_bench_tp${tp}_$rep.err
    python - $OUT/r05_soil_bench_tp${tp}_$rep.json $tp <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d.get("soil", d)
print("two_pass=%s" % sys.argv[2], {k: (v["ms_per_step"], v["frac_hbm"], v["multi_substep_columns_frac"]) for k, v in s.items() if isinstance(v, dict) and "ms_per_step" in v})
PY
  done
done
