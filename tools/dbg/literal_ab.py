#!/usr/bin/env python3
"""A/B of the literal metric (FASTQ -> three sorted BAMs + BAI through the reference's unmodified script, fused hand-off, one GPU) under several
host-side settings: one reference, one index, one FASTQ, then the script once per configuration.  No verification of the BAMs (tools/soak.py does
that); the first configuration runs twice (the first run warms the page cache).
usage: literal_ab.py [--pairs N] CONFIG...      CONFIG = name[:t=THREADS][:ranks=N][:VAR=value...]   e.g.  quota:t=16:SSG_SORT_THREADS=24:SSG_FMT_THREADS=12"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from speedseq_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8000000)
    ap.add_argument("--ref-mbp", type=float, default=3100.0)
    ap.add_argument("--mem", type=int, default=64)
    ap.add_argument("--no-warmup", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "literal_ab.json"))
    ap.add_argument("configs", nargs="+")
    a = ap.parse_args()
    torch.cuda.init(); dev = torch.device("cuda", 0)
    lib = capi.Lib()
    ref, lens, _ = bench.synth_reference(int(a.ref_mbp * 1e6), 20150810, dev)
    names = bench.GRCH37_NAMES[:len(lens)]
    ctg_off = np.concatenate([[0], np.cumsum(lens)])[:-1]
    idx = lib.index_build_dev(ref.data_ptr(), int(ref.numel()), ctg_off, lens, names)
    td_obj = tempfile.TemporaryDirectory(dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None); td = td_obj.name
    prefix = os.path.join(td, "ref.fa")
    lib.index_save(idx, prefix); lib.index_destroy(idx)
    fq = os.path.join(td, "reads.fq")
    done, chunk = 0, 4000000
    with open(fq, "wb") as f:
        while done < a.pairs:
            n = min(chunk, a.pairs - done)
            r = bench.simulate_pairs(ref, lens, n, 150, 1000 + done // chunk, dev).cpu().numpy()
            part = os.path.join(td, "part.fq")
            bench.write_fastq(part, r, 150, first_pair=done)
            f.write(open(part, "rb").read()); os.remove(part)
            done += n
    del ref
    torch.cuda.empty_cache()
    bench.log("index + FASTQ of %d pairs ready; host cpu quota: %s, os.cpu_count: %s" % (a.pairs, bench.host_cpu_quota(), os.cpu_count()))
    b = lambda n: os.path.join(ROOT, "bin", n)
    results = []
    for k, cfg in enumerate(([a.configs[0]] if not a.no_warmup else []) + a.configs):
        parts = cfg.split(":")
        threads, env, ranks, trace = 32, {}, 0, False
        for p in parts[1:]:
            key, _, val = p.partition("=")
            if key == "t":
                threads = int(val)
            elif key == "trace":          # the kernels of `bwa mem` alone, as rocprofv3 sees them inside the pipeline (tools/dbg/trace_util.py prints who waits for whom)
                trace = True
            elif key == "ranks":
                ranks = int(val)
            else:
                env[key] = val
        extra = "export SSG_FUSED=1\nexport SSG_SORT_LOG=1\nexport SSG_STAMP=1\nexport SSG_SBL_LOG=1\n" + "".join("export %s=%s\n" % kv for kv in env.items())
        import resource
        def cpu_stat():
            try:
                return {l.split()[0]: int(l.split()[1]) for l in open("/sys/fs/cgroup/cpu.stat").read().split("\n") if l}
            except Exception:
                return {}
        cs0 = cpu_stat()
        ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
        bwa_cmd = b("bwa")
        if trace:
            tdir = "/tmp/prof_bwa_%d" % k
            bwa_cmd = "env TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d %s -o bwa -- %s" % (tdir, b("bwa"))
        # every stage behind a shell that says when the process was really gone (what the kernel does at exit -- unmapping, the GPU context -- comes after the last line a process can print)
        wrapdir = os.path.join(ROOT, "gpurun_out", ".abwrap_%d" % k); os.makedirs(wrapdir, exist_ok=True)
        smb = os.path.join(wrapdir, "sambamba")
        open(smb, "w").write("#!/bin/sh\n%s \"$@\"; rc=$?; echo \"[stamp] sambamba_$1 exited $(date +%%s.%%N)\" >&2; exit $rc\n" % b("sambamba"))
        os.chmod(smb, 0o755)
        bwa_cmd = "sh -c '%s \"$@\"; rc=$?; echo \"[stamp] bwa exited $(date +%%s.%%N)\" >&2; exit $rc' bwa" % bwa_cmd if False else bwa_cmd
        r = bench.script_leg(td, "ab%d" % k, prefix, fq, a.pairs, threads, bwa_cmd, b("samblaster"), smb, sort_mem_gb=a.mem, config_extra=extra, limit_s=400, ranks=ranks, env_extra={"SSG_RANKS_KEEP_DEVICES": "1"} if ranks > 1 else None)
        for x in (".bam", ".splitters.bam", ".discordants.bam"):
            for y in ("", ".bai"):
                try:
                    os.remove(r.get("out", "") + x + y)
                except OSError:
                    pass
        if trace:
            import subprocess
            tr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dbg", "trace_util.py"), tdir], capture_output=True, text=True)
            open(os.path.join(ROOT, "gpurun_out", "literal_ab_trace_%d.txt" % k), "w").write(tr.stdout + tr.stderr)
            bench.log(tr.stdout[-3000:])
        keep = [l[:260] for l in r.get("stage_log", []) if "[bwa]" in l or "records" in l or "merge" in l or "[samblaster]" in l or "[sambamba] sort: write" in l]
        ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
        cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)   # CPU seconds of every process of the pipeline: against wall x the host's CPU quota
        res = {"config": cfg + (" (warm-up)" if k == 0 and not a.no_warmup else ""), "wall_s": r.get("wall_s"), "children_cpu_s": round(cpu_s, 2), "children_user_s": round(ru1.ru_utime - ru0.ru_utime, 2), "pairs_per_s": round(r.get("pairs_per_s", 0)), "error": r.get("error"), "timeline_s": r.get("timeline_s"), "stage_log": keep, "cgroup_throttled": {k: cpu_stat().get(k, 0) - cs0.get(k, 0) for k in ("nr_periods", "nr_throttled", "throttled_usec")}}
        results.append(res)
        bench.log(json.dumps({k2: v for k2, v in res.items() if k2 != "stage_log"}))
        for l in keep:
            bench.log("      " + l)
        bench.log("      timeline: " + " | ".join(r.get("timeline_s") or []))
    json.dump(results, open(a.out, "w"), indent=1)
    td_obj.cleanup()


if __name__ == "__main__":
    main()
