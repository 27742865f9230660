#!/usr/bin/env python3
"""For every rocprofv3 kernel-trace CSV under a directory (one per traced process): launches, span from first to last launch, device busy time (union over
queues), idle time, and the kernels by summed duration -- which process of a pipeline keeps the device busy, and with what.
usage: trace_busy.py DIR [TOP]"""
import csv
import os
import sys

top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for root, _, files in os.walk(sys.argv[1]):
    for fn in sorted(files):
        if not fn.endswith("kernel_trace.csv"):
            continue
        rows = list(csv.DictReader(open(os.path.join(root, fn))))
        if not rows:
            continue
        iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
        t0, end, busy = iv[0][0], iv[0][0], 0
        for s, e in iv:
            if e > end:
                busy += e - max(s, end)
                end = e
        tot = {}
        for r in rows:
            k = r["Kernel_Name"].split("(")[0][:70]
            a = tot.setdefault(k, [0, 0])
            a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
        print("== %s: %d launches, span %.1f ms, busy %.1f ms, idle %.1f ms" % (fn, len(rows), (end - t0) / 1e6, busy / 1e6, (end - t0 - busy) / 1e6))
        ssum = sum(v[0] for v in tot.values())
        print("   sum of kernel durations %.1f ms (overlap factor %.2f)" % (ssum / 1e6, ssum / max(1, busy)))
        for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:top]:
            print("   %9.1f ms %7d x  %s" % (v[0] / 1e6, v[1], k))
