"""Condenses the rocprofv3 outputs of tools/gpu_profile.sh into one text summary (per kernel: calls,
mean duration, and -- from the PMC passes -- FETCH_SIZE / WRITE_SIZE per launch in bytes).

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 the read counter under-reports wide
coalesced streams (MI355X_MICROARCH.md, HBM section), so the prep kernel -- whose HBM traffic is known
exactly (36 B read + 8 B written per cell, no reuse, working set >> 256 MiB L3) -- is used to calibrate a
correction factor for this engine's 8-byte-per-lane access pattern; corrected numbers are printed next to
the raw ones.
"""
import csv
import glob
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def find(d, pat):
    r = glob.glob(os.path.join(out, d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    import re
    m = re.search(r"(k_level|k_levels_narrow)<([^>]*)>", name)
    if m:   # template flags: FUSED (beta=3/5 quintic), ORDERED (engine-order vectors), INDEXED (row-block partition)
        f = [x.strip() == "true" for x in m.group(2).split(",")] + [False, False, False]
        tag = ("fused" if f[0] else "general") + ("+ordered" if f[1] else "+pixel") + ("+indexed" if f[2] else "")
        return "%s[%s]" % (m.group(1), tag)
    m = re.search(r"(k_fused_cones|k_fused_substeps|k_level_multi|k_levels_narrow_multi|k_sweep_cones)<([^>]*)>", name)
    if m:   # template flags as 0 / 1 (fused sub-steps: SPLIT, ALL35, STRUCT, DIST; sweeps: FUSED, ORDERED, NR)
        return "%s<%s>" % (m.group(1), ",".join("1" if x.strip() == "true" else "0" if x.strip() == "false" else x.strip()
                                                 for x in m.group(2).split(",")))
    m = re.search(r"k_soil_columns<([^>]*)>", name)
    if m:   # FASTPOW, STAGE
        f = [x.strip() == "true" for x in m.group(1).split(",")] + [False, False]
        return "k_soil_columns[%s]" % ("staging" if f[1] else "plain")
    for k in ("k_soil_columns_deferred", "k_accu_cones", "k_jump_init", "k_jump_step", "k_labels",
              "k_take_root", "k_lddrepair", "k_lddmask", "k_parents", "k_downstream", "k_soil_pf", "k_soil_hist", "k_soil_scatter", "k_fused_substeps", "k_canopy", "k_surface_pre",
              "k_surface_post", "k_calib_copy8", "k_calib_copy16", "k_prep", "k_levels_narrow", "k_level", "k_soil_columns", "k_interception", "k_substep",
              "k_halo", "k_gather", "k_scatter"):
        if k in name:
            return k
    return name[:40]


stats = find("prof_%s_kt" % tag, "*kernel_stats.csv")
print("# rocprofv3 --kernel-trace --stats (%s)" % (stats or "missing"))
if stats:
    rows = list(csv.DictReader(open(stats)))
    print("%-34s %8s %14s %14s %8s" % ("kernel", "calls", "total_ms", "mean_us", "pct"))
    for r in rows:
        print("%-34s %8s %14.3f %14.3f %8s" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                               float(r["AverageNs"]) / 1e3, r.get("Percentage", "")))

trace = find("prof_%s_kt" % tag, "*kernel_trace.csv")
cells = {}
if trace:
    per = defaultdict(list)
    for r in csv.DictReader(open(trace)):
        per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                             int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
    print("\n# per kernel from the trace: launches, mean us, mean grid (threads)")
    for k, v in per.items():
        print("%-34s %8d %12.3f %14.1f" % (k, len(v), sum(d for d, _ in v) / len(v) / 1e3,
                                           sum(g for _, g in v) / len(v)))
        cells[k] = sum(g for _, g in v) / len(v)


def pmc(d, counter):
    f = find("prof_%s_%s" % (tag, d), "*counter_collection.csv")
    if not f:
        return {}
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r["Kernel_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
        acc[k][2] += float(r.get("Grid_Size", 0) or 0)
    return acc


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
if fetch or write:
    print("\n# PMC (separate passes), per launch: KiB -> bytes; grid = threads per launch")
    print("%-34s %8s %16s %16s %14s %12s %12s" % ("kernel", "launches", "FETCH_B/launch", "WRITE_B/launch",
                                                   "grid", "fetch_B/thr", "write_B/thr"))
    for k in sorted(set(fetch) | set(write)):
        n = fetch.get(k, write.get(k))[0]
        fb = fetch[k][1] * 1024 / fetch[k][0] if k in fetch else float("nan")
        wb = write[k][1] * 1024 / write[k][0] if k in write else float("nan")
        g = (fetch.get(k) or write.get(k))[2] / n
        print("%-34s %8d %16.0f %16.0f %14.0f %12.2f %12.2f" % (k, n, fb, wb, g, fb / max(g, 1), wb / max(g, 1)))
    for cal in ("k_calib_copy8", "k_calib_copy16"):
        if cal in fetch and cal in write:
            g = fetch[cal][2] / fetch[cal][0]
            per_thr = 8.0 if cal.endswith("8") else 16.0
            cf = per_thr / (fetch[cal][1] * 1024 / fetch[cal][0] / g)
            cw = per_thr / (write[cal][1] * 1024 / write[cal][0] / g)
            print("\n# calibration on %s (exactly %g B read + %g B written per thread): read x%.3f, write x%.3f"
                  % (cal, per_thr, per_thr, cf, cw))
            for k in sorted(set(fetch) & set(write)):
                fb = fetch[k][1] * 1024 / fetch[k][0] * cf
                wb = write[k][1] * 1024 / write[k][0] * cw
                g2 = fetch[k][2] / fetch[k][0]
                print("%-34s corrected HBM bytes/launch: read %.4g + write %.4g = %.4g  (%.1f B per thread)"
                      % (k, fb, wb, fb + wb, (fb + wb) / max(g2, 1)))


# machine-readable digest: per kernel, rocprof mean duration and PMC-corrected HBM bytes per launch
import json
digest = {"tag": tag, "fetch_correction": 2.0, "write_correction": 1.0, "kernels": {}}
if stats:
    for r in csv.DictReader(open(stats)):
        digest["kernels"].setdefault(short(r["Name"]), {}).update(
            calls=int(r["Calls"]), mean_us=float(r["AverageNs"]) / 1e3)
for k in set(fetch) & set(write):
    d = digest["kernels"].setdefault(k, {})
    d["hbm_read_bytes_per_launch"] = fetch[k][1] * 1024 / fetch[k][0] * 2.0
    d["hbm_write_bytes_per_launch"] = write[k][1] * 1024 / write[k][0] * 1.0
    d["threads_per_launch"] = fetch[k][2] / fetch[k][0]
json.dump(digest, open(os.path.join(out, "prof_%s_digest.json" % tag), "w"), indent=1, sort_keys=True)
