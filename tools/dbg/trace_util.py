#!/usr/bin/env python3
"""Device utilisation of one traced process (rocprofv3 --kernel-trace CSV under DIR): busy = union of all launches; per 100 ms bin the busy fraction; per queue the busy
time; the largest idle gaps with the kernels on either side; the kernels by summed time.  usage: trace_util.py DIR"""
import csv
import os
import sys

rows = []
for root, _, files in os.walk(sys.argv[1]):
    for fn in files:
        if fn.endswith("kernel_trace.csv"):
            rows += list(csv.DictReader(open(os.path.join(root, fn))))
if not rows:
    sys.exit("no kernel trace under " + sys.argv[1])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "")) for r in rows)
t0 = iv[0][0]
t1 = max(e for _, e, _, _ in iv)
busy, end, gaps, last = 0, t0, [], iv[0][2]
for s, e, n, q in iv:
    if s > end:
        gaps.append((s - end, (end - t0) / 1e6, last, n))
    if e > end:
        busy += e - max(s, end)
        end, last = e, n
print("%d launches over %.1f ms: device busy %.1f ms (%.0f %%), sum of kernel time %.1f ms" % (len(iv), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), sum(e - s for s, e, _, _ in iv) / 1e6))
# utilisation per 100 ms
B = 100e6
nb = int((t1 - t0) / B) + 1
ub = [0.0] * nb
end = t0
for s, e, n, q in iv:
    a = max(s, end)
    if e > a:
        k = int((a - t0) / B)
        while a < e:
            lim = t0 + (k + 1) * B
            ub[k] += min(e, lim) - a
            a = min(e, lim); k += 1
        end = e
print("busy %% per 100 ms: " + " ".join("%d" % round(100 * x / B) for x in ub))
qs = {}
for s, e, n, q in iv:
    qs.setdefault(q, [0, 0]); qs[q][0] += e - s; qs[q][1] += 1
print("queues: " + ", ".join("%s: %.0f ms in %d" % (q, v[0] / 1e6, v[1]) for q, v in sorted(qs.items())))
hist = {}
for g, at, a, b in gaps:
    k = "<50us" if g < 50e3 else "<200us" if g < 200e3 else "<1ms" if g < 1e6 else "<10ms" if g < 10e6 else ">=10ms"
    h = hist.setdefault(k, [0, 0]); h[0] += 1; h[1] += g
print("idle gaps: " + ", ".join("%s: %d = %.1f ms" % (k, v[0], v[1] / 1e6) for k, v in hist.items()))
pair = {}
for g, at, a, b in gaps:
    h = pair.setdefault((a, b), [0, 0]); h[0] += 1; h[1] += g
print("idle time by (kernel before -> kernel after):")
for (a, b), v in sorted(pair.items(), key=lambda kv: -kv[1][1])[:30]:
    print("  %8.1f ms in %4d gaps  %s -> %s" % (v[1] / 1e6, v[0], a, b))
tot = {}
for s, e, n, q in iv:
    h = tot.setdefault(n, [0, 0]); h[0] += e - s; h[1] += 1
print("kernels by time:")
for n, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:25]:
    print("  %8.1f ms %6d x  %s" % (v[0] / 1e6, v[1], n))
