set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "VALUBusy" "MemUnitBusy" "MemUnitStalled" "VALUUtilization" "SQ_INSTS_VALU" "SQ_WAVES" "SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_soil_$C -o pmc -- python $ROOT/bench.py --only soil --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_soil_$C.err; echo "$C rc=$?"
done
cd $ROOT
python - <<'PY'
import csv, glob, collections, os
out=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out")
for d in sorted(glob.glob(out+"/pmc_soil_*/")):
    f=glob.glob(d+"**/*counter_collection.csv",recursive=True)
    if not f: print(d,"no csv"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        n=r["Kernel_Name"]
        k="deferred" if "deferred" in n else "pass1" if "k_soil_columns" in n else None
        if k: agg[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        print(k, "n=%d"%len(v), " ".join("%.1f"%x for x in v))
PY
find $OUT -name "*.db" -delete
