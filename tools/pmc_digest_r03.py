"""profiles/pmc_<tag>_<workload>.txt (tools/pmc_r03.sh) -> profiles/<tag>_pmc_digest.json: per workload and kernel the
launches, HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes; the x2 is the gfx950 calibration of
DESIGN.md section 5) and the SQ counters per wavefront, for bench.py's roofline.traffic.
    python tools/pmc_digest_r03.py r03a"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03a"
out = {"tag": tag, "fetch_correction": 2.0, "workloads": {}}
for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_%s_*.txt" % tag))):
    wl = os.path.basename(f)[len("pmc_%s_" % tag):-4]
    kernels = {}
    for line in open(f):
        if line.startswith("#") or "launches=" not in line:
            continue
        name = line[:line.index("launches=")].strip()
        vals = dict(re.findall(r"(\w+)=([0-9.e+]+)", line))
        n = int(float(vals["launches"]))
        k = {"launches": n}
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            k["hbm_read_bytes_per_launch"] = 2.0 * float(vals["FETCH_SIZE"]) * 1024 / n
            k["hbm_write_bytes_per_launch"] = float(vals["WRITE_SIZE"]) * 1024 / n
        if "SQ_WAVES" in vals and float(vals["SQ_WAVES"]) > 0:
            w = float(vals["SQ_WAVES"])
            k["waves_per_launch"] = w / n
            for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                      "SQ_INSTS_VALU"):
                if c in vals:
                    k[c.lower() + "_per_wave"] = float(vals[c]) / w
            if "GRBM_GUI_ACTIVE" in vals:
                k["gui_active_cycles_per_launch_per_xcd"] = float(vals["GRBM_GUI_ACTIVE"]) / 8 / n
        kernels[name] = k
    out["workloads"][wl] = {"source": os.path.relpath(f, ROOT), "kernels": kernels}
dst = os.path.join(ROOT, "profiles", "%s_pmc_digest.json" % tag)
json.dump(out, open(dst, "w"), indent=1)
print(dst)
