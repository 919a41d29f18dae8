"""Same-process-free A/B of two builds of the library on the headline call: python tools/ab_two_libs.py libA.so libB.so [reps]
(alternates subprocesses of tools/bench_headline_ab.py with LISFLOOD_AMD_LIBRARY set)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for rep in range(reps):
    for lib in libs:
        env = dict(os.environ)
        if lib != "default":
            env["LISFLOOD_AMD_LIBRARY"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_headline_ab.py"), "10000", "30"], env=env,
                             capture_output=True, text=True).stdout
        print(lib, "|", " | ".join(l.split(": ", 1)[1] for l in out.strip().splitlines() if ": " in l), flush=True)
