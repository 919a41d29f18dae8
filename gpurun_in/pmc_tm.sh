cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_tm
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_tm -o pmc -- python $GRAFT_REPO_ROOT/bench.py --only model_step --size 6000 --family shallow > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/pmc_tm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"\(.*", "", k)
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, c in agg.items():
    if "level_steps" not in k: continue
    n = max(v[0] for v in c.values()); w = c["SQ_WAVES"][1]
    print(k, "launches", n, {cn: round(v[1] / w, 1) for cn, v in c.items() if cn != "SQ_WAVES"}, "waves", w)
    print("  VALU-active share of wave cycles: %.3f; wait_any share %.3f; gui cycles per launch per xcd %.0f" % (
        c["SQ_ACTIVE_INST_VALU"][1] * 4 / c["SQ_WAVE_CYCLES"][1] if False else c["SQ_ACTIVE_INST_VALU"][1] / c["SQ_WAVE_CYCLES"][1],
        c["SQ_WAIT_ANY"][1] / c["SQ_WAVE_CYCLES"][1], c["GRBM_GUI_ACTIVE"][1] / 8 / n))
    print("  busy: SQ_ACTIVE_INST_VALU / (GUI cycles per xcd * SIMDs per xcd 128): %.3f" % (c["SQ_ACTIVE_INST_VALU"][1] / (c["GRBM_GUI_ACTIVE"][1] / 8 * 128 * 8)))
PY
