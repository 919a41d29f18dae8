"""Same-box A/B of the headline call between two builds of the library: python tools/bench_headline_ab.py [size] [steps]
(run once per LISFLOOD_AMD_LIBRARY setting; prints ms per call and the wide-level kernel mean)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
import bench  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
kw, p, g = bench.build_case("shallow", size, size)
for rep in range(3):
    r = bench.run_routing(kw, p, steps, 3)
    w = r["prof"]["wide_level"]
    us = w["ms"] * 1e3 / max(w["launches"], 1)
    cells = w["cells"] / max(w["launches"], 1)
    print("%s: %.4f ms per call  %.1f Gcell-steps/s  wide level mean %.1f us -> frac %.4f" % (
        os.path.basename(os.environ.get("LISFLOOD_AMD_LIBRARY", "default")), r["ms_per_step"],
        kw.num_pixels / r["ms_per_step"] / 1e6, us, 48.0 * cells / (us * 1e-6) / 8e12), flush=True)
