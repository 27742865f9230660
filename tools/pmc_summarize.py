#!/usr/bin/env python3
"""Summaries of a round's rocprofv3 counter passes (tools/profile_round.sh) -> profiles/<tag>_pmc_traffic.json and
profiles/<tag>_pmc_sq.json.  HBM traffic: FETCH_SIZE / WRITE_SIZE per launch, with the unit calibrated on the random-gather
probe of the same session (known number of 64-byte lines per launch).  Issue counters: VALU instructions per launch ->
lane-operations per second against the measured int32 VALU probe."""
import csv
import json
import re
import sys
from collections import defaultdict


def load(path):
    rows = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> values per dispatch
    dur = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    return rows, dur


def main():
    tag, outdir, src = sys.argv[1], sys.argv[2], sys.argv[3]
    # a workload tag (r06_cfg5) takes the probes and their calibration passes from its round's own profile (r06), here or under profiles/
    import os
    base = tag.split("_")[0]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def aux(name):
        for t, d in ((tag, src), (base, src), (base, os.path.join(root, "profiles"))):
            q = "%s/%s_%s" % (d, t, name)
            if os.path.exists(q):
                return q
        import glob
        older = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_" + name)))   # the newest earlier round's (same part, same probes)
        return older[-1] if older else "%s/%s_%s" % (src, tag, name)
    workload = {"pairs": 200000, "read_len": 250} if "cfg5" in tag else {"pairs": 1000000, "read_len": 150}
    # ---- calibration: the probe reads lanes * iters * lines_per_step random 64-byte lines per launch ----
    cal, _ = load(aux("pmc_probe_calibration.csv"))
    lines = {"probe<0>": 1, "probe<1>": 1, "probe<2>": 2, "probe<3>": 0.25, "probe<4>": 0.5}
    ratios = []
    for k, c in cal.items():
        if k not in lines:
            continue
        for v, n in zip(c["FETCH_SIZE"], [(262144, 50), (262144, 400), (524288, 50), (524288, 400)] * 8):
            exp = n[0] * n[1] * lines[k] * 64.0
            if n[1] == 400:
                ratios.append(v / exp)
    unit = 1.0 / (sum(ratios) / len(ratios))                 # bytes per counter unit for random 64-byte line fetches
    # streaming calibration (tools/dbg/gather_probe.cpp stream_read / stream_write: 4 GiB of 16-byte-per-lane coalesced accesses per launch): the
    # microarchitecture guide says FETCH_SIZE shows HALF the bytes of wide coalesced reads on gfx950 -- measured here, per counter
    STREAM_BYTES = 4.0 * (1 << 30)
    unit_stream_read = unit_stream_write = None
    sr = cal.get("stream_read", {}).get("FETCH_SIZE", [])
    if sr:
        unit_stream_read = STREAM_BYTES / (sum(sr) / len(sr))
    try:
        calw, _ = load(aux("pmc_probe_calibration_write.csv"))
        sw = calw.get("stream_write", {}).get("WRITE_SIZE", [])
        if sw:
            unit_stream_write = STREAM_BYTES / (sum(sw) / len(sw))
    except OSError:
        pass
    fetch, _ = load("%s/%s_pmc_FETCH_SIZE.csv" % (src, tag))
    write, _ = load("%s/%s_pmc_WRITE_SIZE.csv" % (src, tag))
    traffic = {}
    for k in fetch:
        f = fetch[k]["FETCH_SIZE"]; w = write.get(k, {}).get("WRITE_SIZE", [0.0])
        traffic[k] = {"launches": len(f), "fetch_bytes_per_launch": sum(f) / len(f) * unit, "write_bytes_per_launch": sum(w) / max(1, len(w)) * (unit_stream_write or unit),
                      # the same counts priced as coalesced streaming reads: an upper bound for kernels that read arrays in order
                      "fetch_bytes_per_launch_if_streaming": sum(f) / len(f) * unit_stream_read if unit_stream_read else None}
    json.dump({"tag": tag, "pairs": workload["pairs"], "read_len": workload["read_len"], "ref_mbp": 3100.0,
               "calibration": {"bytes_per_counter_unit": unit, "bytes_per_counter_unit_streaming_read": unit_stream_read, "bytes_per_counter_unit_streaming_write": unit_stream_write, "probe_launches_used": len(ratios), "spread": [min(ratios) * unit, max(ratios) * unit],
                               "how": "tools/dbg/gather_probe under --pmc FETCH_SIZE: launches with a known count of random 64-byte line reads; WRITE_SIZE priced with the streaming-write unit when the probe's write pass is there; fetch also priced as coalesced streaming reads (an upper bound for in-order readers)"},
               "bytes_per_launch": {k: v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"] for k, v in traffic.items()},
               "detail": traffic}, open("%s/%s_pmc_traffic.json" % (outdir, tag), "w"), indent=1)
    # ---- issue counters ----
    sq, dur = load("%s/%s_pmc_SQ_INSTS_VALU.csv" % (src, tag))
    sq2, _ = load("%s/%s_pmc_SQ_INSTS_SALU.csv" % (src, tag))
    valu_peak = None
    try:
        for line in open(aux("valu_probe.txt")):
            m = re.search(r"v_add_u32 \+ v_max_i32\s+waves/CU\s+16.0\s+\S+ ms\s+(\S+) T lane-ops/s", line)
            if m:
                valu_peak = float(m.group(1)) * 1e12
    except OSError:
        pass
    stats = {}                                              # un-instrumented durations: the --kernel-trace --stats pass of the same command
    with open("%s/%s_kernel_stats.csv" % (src, tag)) as f:
        for r in csv.DictReader(f):
            stats[re.sub(r"^void ", "", r["Name"]).split("(")[0]] = float(r["AverageNs"]) * 1e-6
    out = {}
    for k in sq:
        c = sq[k]; n = len(c["SQ_INSTS_VALU"]); ms = stats.get(k, sum(dur[k]) / n)
        row = {"launches": n, "ms_per_launch": ms, "ms_per_launch_under_pmc": sum(dur[k]) / n}
        for name in c:
            row[name + "_per_launch"] = sum(c[name]) / n
        for name in sq2.get(k, {}):
            row[name + "_per_launch"] = sum(sq2[k][name]) / len(sq2[k][name])
        row["valu_lane_ops_per_s"] = row["SQ_INSTS_VALU_per_launch"] * 64 / (ms * 1e-3)
        if valu_peak:
            row["valu_frac_of_probe_peak"] = row["valu_lane_ops_per_s"] / valu_peak
        out[k] = row
    json.dump({"tag": tag, "workload": workload, "valu_probe_peak_lane_ops_per_s": valu_peak,
               "note": "SQ_INSTS_VALU counts wave64 instructions; lane-ops = x 64 (inactive lanes included); peak = tools/dbg/valu_probe, add + max mix at 16 waves/CU",
               "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["ms_per_launch"] * kv[1]["launches"]))},
              open("%s/%s_pmc_sq.json" % (outdir, tag), "w"), indent=1)
    for k, v in list(json.load(open("%s/%s_pmc_sq.json" % (outdir, tag)))["kernels"].items())[:8]:
        t = traffic.get(k, {})
        print("%-28s %7.1f ms  VALU %.2e inst  frac %.2f  fetch %.1f GB  write %.1f GB" % (k, v["ms_per_launch"], v["SQ_INSTS_VALU_per_launch"], v.get("valu_frac_of_probe_peak", 0),
              t.get("fetch_bytes_per_launch", 0) / 1e9, t.get("write_bytes_per_launch", 0) / 1e9))


if __name__ == "__main__":
    main()
