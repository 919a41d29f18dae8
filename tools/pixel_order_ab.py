"""The literal drop-in call on device vectors in the reference's pixel order (lf_router_route_device), ms per call:
python tools/pixel_order_ab.py [size] [steps]   (run once per LISFLOOD_AMD_LIBRARY; tools/ab_two_libs.py-style alternation by hand)"""
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
    r = bench.run_routing(kw, p, steps, 3, ordered=False)
    print("%s: pixel-order call %.4f ms  crc %08x" % (os.path.basename(os.environ.get("LISFLOOD_AMD_LIBRARY", "default")), r["ms_per_step"], r["checksum"]), flush=True)
