#!/bin/bash
# same-box A/B: default library (3 fp64 Newton steps in lf_solve_3_5) vs liblisflood_alt.so (-DLF_SOLVE_F64_STEPS=2)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ALT=$ROOT/lisflood-code_amd/lisflood_amd/liblisflood_alt.so
cd $ROOT
for fam in deep river shallow; do
  for rep in 1 2; do
    python tools/ab_route_dump.py $fam 5000 /tmp/q_def_$fam.npy
    LISFLOOD_AMD_LIBRARY=$ALT python tools/ab_route_dump.py $fam 5000 /tmp/q_alt_$fam.npy
  done
  python - <<PY
import numpy as np
a=np.load("/tmp/q_def_$fam.npy"); b=np.load("/tmp/q_alt_$fam.npy")
d=np.abs(a-b)/np.maximum(np.abs(a),1e-300)
print("$fam: max rel diff %.3e, cells differing %d of %d" % (d.max(), int((a!=b).sum()), a.size))
PY
done
for rep in 1 2; do
python bench.py --only model_step 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default fused model step', json.dumps(d)[:300])"
LISFLOOD_AMD_LIBRARY=$ALT python bench.py --only model_step 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('alt     fused model step', json.dumps(d)[:300])"
done
