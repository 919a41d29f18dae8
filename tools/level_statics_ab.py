"""Same-process A/B of the wide-level kernel's static streams: python tools/level_adx_ab.py [size] [steps] [family]
LF_LEVEL_STATICS=0: a and dx as two loads (rounds 1-5); 1 / unset: one (a, dx) record"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
import bench  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
family = sys.argv[3] if len(sys.argv) > 3 else "shallow"
kw, p, g = bench.build_case(family, size, size)
for x in (sys.argv[4].split(",") if len(sys.argv) > 4 else ("0", "1", "0", "1", "0", "1")):
    os.environ["LF_LEVEL_STATICS"] = x
    out = []
    for rep in range(2):
        r = bench.run_routing(kw, p, steps, 3)
        w = r["prof"]["wide_level"]
        out.append("%.4f (wide level %.1f us)" % (r["ms_per_step"], w["ms"] * 1e3 / max(w["launches"], 1)))
    print("LF_LEVEL_STATICS=%s ms per call: %s crc %08x" % (x, "  ".join(out), r["checksum"]), flush=True)
