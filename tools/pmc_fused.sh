#!/bin/bash
# PMC counters of the fused sub-step kernels on deep 5000^2 (separate passes, one counter each)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU" "SQ_WAVES" "SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY" "SQ_INSTS_SALU" "SQ_WAIT_ANY" "SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pf_$C
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pf_$C -o pmc -- python $ROOT/tools/bench_fused_levels.py deep ${1:-3000} ${2:-16} > /dev/null 2>&1; echo "$C rc=$?"
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("/tmp/pf_*/")):
    f=glob.glob(d+"**/*counter_collection.csv",recursive=True)
    if not f: print(d,"no csv"); continue
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f[0])):
        n=r["Kernel_Name"]
        k="cones" if "k_fused_cones" in n else "levels" if "k_fused_substeps" in n else None
        if k:
            a=agg[(k,r["Counter_Name"])]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,v in sorted(agg.items()):
        print(k, "launches=%d total=%.4g per_launch=%.4g"%(v[0],v[1],v[1]/v[0]))
PY
