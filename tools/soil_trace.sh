#!/bin/bash
# per-launch durations (us) of the soil kernels of `bench.py --only soil`, in launch order, for each library given
# usage: tools/soil_trace.sh [path/to/lib.so ...]      ("default" = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for LIB in "$@"; do
  rm -rf /tmp/st
  if [ "$LIB" = default ]; then unset LISFLOOD_AMD_LIBRARY; else export LISFLOOD_AMD_LIBRARY=$ROOT/$LIB; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o kt -- python $ROOT/bench.py --only soil > /tmp/st.json 2>/dev/null
  python - "$LIB" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/st/**/kt_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
p1 = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_soil_columns<" in r["Kernel_Name"]]
p2 = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "deferred" in r["Kernel_Name"]]
print(sys.argv[1]); print("  pass1", " ".join("%.0f" % x for x in p1[:12])); print("  pass2", " ".join("%.0f" % x for x in p2[:12]))
PY
done
