#!/usr/bin/env python3
"""Idle time of the device inside the last step of a rocprofv3 --kernel-trace CSV: the step runs from its first seeding kernel to the last launch of the trace;
busy = union of all launches over all queues.  Prints the total, the busy time, and the largest gaps with the launches on either side (a gap is a host round
trip: a count read back, an allocation, a sort's size query).  usage: step_gaps.py TRACE.csv [N_GAPS]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "ssg_k_smem2" in r["Kernel_Name"] or "ssg_k_smem_kt" in r["Kernel_Name"]]
if not starts:
    sys.exit("no seeding kernel in the trace")
# steps are separated by long stretches without a seeding kernel: take the last run of launches that begins with one
i0 = starts[-1]
while i0 > 0 and any(("ssg_k_smem" in rows[j]["Kernel_Name"]) for j in range(max(0, i0 - 6), i0)):
    i0 = [j for j in range(max(0, i0 - 6), i0) if "ssg_k_smem" in rows[j]["Kernel_Name"]][0]
step = rows[i0:]
t0 = int(step[0]["Start_Timestamp"])
end = t0
busy = 0
gaps = []
prev = step[0]
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > end:
        gaps.append((s - end, (end - t0) / 1e6, prev["Kernel_Name"][:50], r["Kernel_Name"][:50]))
        busy += 0
    if e > end:
        busy += e - max(s, end)
        end = e
        prev = r
total = end - t0
print("step %.1f ms, device busy %.1f ms, idle %.1f ms in %d gaps" % (total / 1e6, busy / 1e6, (total - busy) / 1e6, len(gaps)))
for g, at, a, b in sorted(gaps, reverse=True)[:top]:
    print("%7.3f ms idle at %8.2f ms  after %-50s before %s" % (g / 1e6, at, a, b))
