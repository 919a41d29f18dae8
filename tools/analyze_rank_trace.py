import csv, glob, sys, re, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "kernels")
# take the last third (timed region steady state)
t0 = int(rows[0]["Start_Timestamp"])
ev = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:40], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
n = len(ev)
seg = ev[int(n * 0.6):]
busy = sum(e - s for s, e, _, _ in seg)
span = seg[-1][1] - seg[0][0]
print("last 40%% of kernels: span %.3f ms, sum of kernel durations %.3f ms" % (span / 1e6, busy / 1e6))
by = collections.defaultdict(lambda: [0, 0])
for s, e, k, q in seg:
    by[(k, q)][0] += 1; by[(k, q)][1] += e - s
for (k, q), (c, d) in sorted(by.items(), key=lambda x: -x[1][1])[:12]:
    print("%-42s queue %-6s calls %5d total %.3f ms mean %.1f us" % (k, q, c, d / 1e6, d / c / 1e3))
print("--- a stretch of consecutive kernels (start us, dur us, gap to previous end us):")
prev = seg[0][0]
for s, e, k, q in seg[:70]:
    print("%10.1f %8.1f %8.1f  q%-4s %s" % ((s - seg[0][0]) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, q, k))
    prev = e
