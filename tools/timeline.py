#!/usr/bin/env python3
"""Timeline of the last bench step from a rocprofv3 --kernel-trace CSV: wall, union of kernel-busy time, idle gaps with their
neighbours, per-stream busy time.  usage: tools/timeline.py <kernel_trace.csv> [out.txt]"""
import csv, sys, re

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
short = lambda n: re.sub(r"\(.*", "", n)[:60]
big = [i for i, r in enumerate(rows) if "ssg_k_smem2" in r[2] and r[1] - r[0] > 20e6]
i0 = big[-1]
# the step starts a few small kernels before the SMEM launch (pestat reset etc.): walk back while gaps are < 50 us
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 50e3 and "ssg_k" in rows[i0 - 1][2]:
    i0 -= 1
step = rows[i0:]
t0, t1 = step[0][0], max(r[1] for r in step)
out = []
out.append("last step: %d launches, wall %.2f ms" % (len(step), (t1 - t0) / 1e6))
ev = sorted((r[0], r[1], r[2]) for r in step)
busy, cur_s, cur_e, gaps, last_name = 0, ev[0][0], ev[0][1], [], ev[0][2]
for s, e, n in ev[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, last_name, n, cur_e - t0)); busy += cur_e - cur_s; cur_s, cur_e, last_name = s, e, n
    elif e > cur_e:
        cur_e, last_name = e, n
busy += cur_e - cur_s
out.append("busy (union of kernels) %.2f ms, idle %.2f ms in %d gaps" % (busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps)))
gaps.sort(reverse=True)
out.append("largest gaps (ms, at ms, after -> before):")
for g, a, b, at in gaps[:40]:
    out.append("  %.3f  @%.1f  %s -> %s" % (g / 1e6, at / 1e6, short(a), short(b)))
hist = {}
for g, a, b, at in gaps:
    k = "<20us" if g < 20e3 else "<100us" if g < 100e3 else "<500us" if g < 500e3 else ">=500us"
    h = hist.setdefault(k, [0, 0]); h[0] += 1; h[1] += g
out.append("gap histogram: " + ", ".join("%s: %d gaps %.2f ms" % (k, v[0], v[1] / 1e6) for k, v in hist.items()))
# phases: time from the start of one big kernel to the next
out.append("kernels > 2 ms (start ms, dur ms, queue):")
for s, e, n, q, st in step:
    if e - s > 2e6:
        out.append("  %8.2f %8.2f  q%s s%s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, st, short(n)))
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
