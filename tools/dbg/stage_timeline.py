#!/usr/bin/env python3
"""Timeline of one device step from a rocprofv3 --kernel-trace CSV: every kernel launch between two named kernels (default: the chaining stage, from
ssg_k_sal to ssg_k_ext_prep) with its start and end relative to the first one's end, in ms.  usage: stage_timeline.py TRACE.csv [FIRST [LAST]]"""
import csv
import sys

path = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "ssg_k_sal("
last = sys.argv[3] if len(sys.argv) > 3 else "ssg_k_ext_prep"
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST occurrence of `first` that is followed by `last` (the last step of the run)
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
if not idx:
    sys.exit("no kernel named like %r" % first)
i0 = idx[-1]
t0 = int(rows[i0]["End_Timestamp"])
print("t = 0 at the end of %s" % rows[i0]["Kernel_Name"][:60])
for r in rows[i0 + 1:]:
    name = r["Kernel_Name"]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print("%8.3f %8.3f  %7.3f  q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), name[:90]))
    if last in name:
        break
