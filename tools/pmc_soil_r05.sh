#!/bin/bash
# Round-5 counter passes of the one-launch soil kernel (k_soil_fused) on the GPU box: three SEPARATE rocprofv3 --pmc passes
# (FETCH_SIZE | WRITE_SIZE | SQ + GRBM) of `LF_BENCH_SOIL_REGIME=<regime> bench.py --only soil`, condensed per kernel into
# gpurun_out/pmc_<tag>_soil_<regime>_4000000.txt (format of tools/pmc_r03.sh; tools/pmc_digest_r03.py reads it).
# usage: pmc_soil_r05.sh <tag> [passes] [regimes]     e.g. pmc_soil_r05.sh r05a "SQ" "wet"
set -u
TAG=${1:-r05}; PASSES=${2:-"FETCH_SIZE WRITE_SIZE SQ"}; REGIMES=${3:-"wet single_substep"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS"
for regime in $REGIMES; do
name=soil_${regime}_4000000
for pass in $PASSES; do
  C=$pass; [ $pass = SQ ] && C=$SQ; [ $pass = SQ2 ] && C=$SQ2
  rm -rf /tmp/pmc_${name}_$pass
  LF_BENCH_SOIL_REGIME=$regime timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_${name}_$pass -o pmc -- \
      python $ROOT/bench.py --only soil > /tmp/pmc_${name}_$pass.log 2>&1
  echo "$name $pass rc=$?"
done
python - "$name" "$PASSES" > $OUT/pmc_${TAG}_${name}.txt <<'PY'
import csv, glob, collections, re, sys
name, passes = sys.argv[1], sys.argv[2].split()
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for p in passes:
    for f in glob.glob("/tmp/pmc_%s_%s/**/*counter_collection.csv" % (name, p), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"\(.*", "", k)
            a = agg[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
print("# %s: per-kernel counter totals over `bench.py --only soil` of that regime (12 calls; rocprofv3 --pmc, separate passes); FETCH/WRITE in KiB as reported" % name)
for k, c in sorted(agg.items()):
    n = max(v[0] for v in c.values())
    line = ["%-44s launches=%d" % (k, n)]
    for cn, v in sorted(c.items()):
        line.append("%s=%.6g" % (cn, v[1]))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        line.append("hbm_bytes_per_launch(read x2 + write)=%.6g" % ((2 * c["FETCH_SIZE"][1] + c["WRITE_SIZE"][1]) * 1024 / n))
    print("  ".join(line))
PY
cat $OUT/pmc_${TAG}_${name}.txt
done
