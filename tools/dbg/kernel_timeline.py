#!/usr/bin/env python3
"""What is on the critical path of the device step: from a rocprofv3 --kernel-trace CSV, the last step (from the last start of the seeding kernel to the last
kernel's end) as (a) per kernel: launches, busy ms (union of its intervals), ms during which nothing else ran ("alone"), (b) the step cut into stretches by the
set of kernels running, merged to a readable list.  usage: kernel_timeline.py TRACE.csv [--first KERNEL_SUBSTRING] [--min-ms 0.5]"""
import csv, sys, argparse, json
ap = argparse.ArgumentParser()
ap.add_argument("trace"); ap.add_argument("--first", default="ssg_k_smem2"); ap.add_argument("--min-ms", type=float, default=0.5); ap.add_argument("--json", default=None)
a = ap.parse_args()
rows = []
with open(a.trace, newline="") as f:
    for r in csv.DictReader(f):
        n = r.get("Kernel_Name") or r.get("Name") or ""
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void ", "").strip()))
rows.sort()
starts = [s for s, e, n in rows if a.first in n]
if not starts: sys.exit("no kernel named like %r in the trace" % a.first)
# the seeding kernel may be launched more than once per step (light + heavy): a new step starts where the gap to the previous seeding launch exceeds 50 ms
cut = starts[-1]
for s in reversed(starts):
    if cut - s < 50e6: cut = s
    else: break
step = [(s, e, n) for s, e, n in rows if s >= cut]
t0 = step[0][0]; t1 = max(e for s, e, n in step)
ev = sorted([(s, 1, n) for s, e, n in step] + [(e, -1, n) for s, e, n in step])
running = {}; last = t0; alone = {}; busy = {}; segs = []
for t, d, n in ev:
    if t > last and running:
        key = tuple(sorted(running))
        for k in key: busy[k] = busy.get(k, 0) + (t - last)
        if len(key) == 1: alone[key[0]] = alone.get(key[0], 0) + (t - last)
        if segs and segs[-1][2] == key: segs[-1][1] = t
        else: segs.append([last, t, key])
    elif t > last and not running:
        if segs and segs[-1][2] == (): segs[-1][1] = t
        else: segs.append([last, t, ()])
    last = t
    running[n] = running.get(n, 0) + d
    if running[n] == 0: del running[n]
ms = lambda x: x / 1e6
print("step: %.1f ms, %d launches" % (ms(t1 - t0), len(step)))
cnt = {}
for s, e, n in step: cnt[n] = cnt.get(n, 0) + 1
print("%-44s %6s %9s %9s" % ("kernel", "n", "busy ms", "alone ms"))
for n in sorted(busy, key=lambda k: -alone.get(k, 0))[:30]:
    print("%-44s %6d %9.2f %9.2f" % (n[:44], cnt[n], ms(busy[n]), ms(alone.get(n, 0))))
idle = sum(e - s for s, e, k in segs if k == ())
print("idle (no kernel running): %.2f ms; alone total %.2f ms" % (ms(idle), ms(sum(alone.values()))))
print("stretches of %.1f ms and more:" % a.min_ms)
for s, e, k in segs:
    if ms(e - s) >= a.min_ms: print("  %8.2f .. %8.2f  (%6.2f)  %s" % (ms(s - t0), ms(e - t0), ms(e - s), ", ".join(x.replace("ssg_k_", "") for x in k) or "-- idle --"))
if a.json:
    json.dump({"step_ms": ms(t1 - t0), "busy_ms": {k: ms(v) for k, v in busy.items()}, "alone_ms": {k: ms(v) for k, v in alone.items()}, "idle_ms": ms(idle)}, open(a.json, "w"), indent=1)
