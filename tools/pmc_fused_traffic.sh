#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of the fused sub-step kernels on deep N^2: HBM bytes per (cell, sub-step)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
SIZE=${1:-3000}
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pft_$C
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pft_$C -o pmc -- python $ROOT/tools/bench_fused_levels.py deep $SIZE 16 > /dev/null 2>&1; echo "$C rc=$?"
done
python - "$SIZE" <<'PY'
import csv, glob, collections, sys
size = int(sys.argv[1]); cells = size * size
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pft_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = "cones" if "k_fused_cones" in r["Kernel_Name"] else "levels" if "k_fused_substeps" in r["Kernel_Name"] else None
        if k:
            tot[(k, c)] += float(r["Counter_Value"]) * 1024 * (2.0 if c == "FETCH_SIZE" else 1.0)   # KiB; gfx950 read correction x2
            n[(k, c)] += 1
# bench_fused_levels runs 4 model steps per configuration (1 + 3 timed), 24 split sub-steps each
for k in ("levels", "cones"):
    rd, wr = tot[(k, "FETCH_SIZE")], tot[(k, "WRITE_SIZE")]
    per = (rd + wr) / (4 * 24 * cells)
    print("%-6s launches %6d  read %.3f GB  written %.3f GB per model step -> %.1f B of HBM traffic per (cell, sub-step)"
          % (k, n[(k, "FETCH_SIZE")], rd / 4 / 1e9, wr / 4 / 1e9, per))
PY
