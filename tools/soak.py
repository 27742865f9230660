#!/usr/bin/env python3
"""Config-3 soak (BASELINE.json configs[2], scaled): tens of millions of synthetic 2x150 pairs vs the 3.1 Gbp GRCh37-shaped reference through
the reference's `speedseq align` script (unmodified) on the product executables, fused hand-off, ONE GPU.  Reports the rate, the peak RSS of
the pipeline's processes, the duplicate-table size, sort spills, and checks samtools-flagstat-level invariants of the three BAMs against what
bwa / samblaster reported on stderr.  Usage: python tools/soak.py --pairs 40000000 [--mem 64] [--emu-selftest]"""
import argparse
import json
import os
import re
import resource
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=40000000)
    ap.add_argument("--chunk", type=int, default=4000000, help="pairs simulated and written per round")
    ap.add_argument("--ref-mbp", type=float, default=3100.0)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--mem", type=int, default=64, help="-M of speedseq align (GB): sambamba sort gets M-2")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--emu-selftest", action="store_true")
    a = ap.parse_args()
    emu = a.emu_selftest
    from speedseq_amd import capi
    import ctypes as C
    if emu:
        import simreads
        dev = torch.device("cpu")
        lib = capi.Lib(os.path.join(ROOT, "tests", "emu", "libssgpu_emu.so"))
        codes = np.concatenate([c for _, c in simreads.read_fasta(os.path.join(ROOT, "tests", "golden", "chr20_slice.fa"))]).astype(np.uint8)
        codes[codes > 3] = 0
        ref, lens = torch.from_numpy(codes), [int(codes.size)]
        b = lambda n: os.path.join(ROOT, "tests", "emu", n + "_emu")
    else:
        torch.cuda.init(); dev = torch.device("cuda", 0)
        lib = capi.Lib()
        ref, lens, _ = bench.synth_reference(int(a.ref_mbp * 1e6), 20150810, dev)
        b = lambda n: os.path.join(ROOT, "bin", n)
    names = bench.GRCH37_NAMES[:len(lens)]
    ctg_off = np.concatenate([[0], np.cumsum(lens)])[:-1]
    t0 = time.time()
    idx = lib.index_build_dev(ref.data_ptr(), int(ref.numel()), ctg_off, lens, names)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    td_obj = tempfile.TemporaryDirectory(dir=shm); td = td_obj.name
    prefix = os.path.join(td, "ref.fa")
    lib.index_save(idx, prefix)
    lib.index_destroy(idx)
    bench.log("reference + index files ready (%.1f s)" % (time.time() - t0))
    fq = os.path.join(td, "reads.fq")
    done = 0
    with open(fq, "wb") as f:
        while done < a.pairs:
            n = min(a.chunk, a.pairs - done)
            r = bench.simulate_pairs(ref, lens, n, a.read_len, 1000 + done // a.chunk, dev).cpu().numpy()
            part = os.path.join(td, "part.fq")
            bench.write_fastq(part, r, a.read_len, first_pair=done)
            with open(part, "rb") as g:
                while True:
                    blk = g.read(64 << 20)
                    if not blk:
                        break
                    f.write(blk)
            os.remove(part)
            done += n
    del ref
    if not emu:
        torch.cuda.empty_cache()
    bench.log("FASTQ of %d pairs written (%.1f GB)" % (a.pairs, os.path.getsize(fq) / 1e9))
    cfg = "export SSG_FUSED=1\nexport SSG_SORT_THREADS=%d\nexport SSG_SORT_LOG=1\n" % min(os.cpu_count() or 8, 128)
    r = bench.script_leg(td, "soak", prefix, fq, a.pairs, a.threads, b("bwa"), b("samblaster"), b("sambamba"), sort_mem_gb=a.mem, config_extra=cfg, limit_s=1500)
    rss_gb = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1048576.0       # largest RSS of any child so far: the pipeline's heaviest process
    out = {"what": "`speedseq align -t %d -M %d -p` (reference script, unmodified; SSG_FUSED=1) on bin/bwa, bin/samblaster, bin/sambamba: %d synthetic 2x%d pairs vs the %.0f Mbp reference, one GPU"
                   % (a.threads, a.mem, a.pairs, a.read_len, sum(lens) / 1e6), "peak_child_rss_gb": round(rss_gb, 2)}
    out.update({k: v for k, v in r.items() if k != "out"})
    if "out" in r:
        samtools = os.path.join(ROOT, "oracle", "_ref", "samtools")
        log = "\n".join(r.get("stage_log", []))
        m = re.search(r"pairs=(\d+) dups=(\d+) discordant_pairs=(\d+) splitter_lines=(\d+)", log)
        rep = dict(zip(("pairs", "dups", "discordant_pairs", "splitter_lines"), map(int, m.groups()))) if m else {}
        out["samblaster_reported"] = rep
        if os.path.exists(samtools):
            from concurrent.futures import ThreadPoolExecutor   # the five passes over the BAMs side by side: each is one samtools thread
            def count(bam, *flt):
                return int(subprocess.check_output([samtools, "view", "-c"] + list(flt) + [bam]))
            main_bam = r["out"] + ".bam"
            with ThreadPoolExecutor(5) as ex:
                f = {"primary_records": ex.submit(count, main_bam, "-F", "0x900"), "dup_flagged_primaries": ex.submit(count, main_bam, "-f", "0x400", "-F", "0x900"),
                     "discordant_records": ex.submit(count, r["out"] + ".discordants.bam"), "splitter_records": ex.submit(count, r["out"] + ".splitters.bam")}
                srt_f = ex.submit(subprocess.run, "%s view %s | cut -f3,4 | awk 'BEGIN{ok=1} { if ($1==c && $2<p) ok=0; c=$1; p=$2 } END{print ok}'" % (samtools, main_bam), shell=True, capture_output=True, text=True)
                chk = {k: v.result() for k, v in f.items()}
                srt = srt_f.result()
            out["bam_counts"] = chk
            ok = chk["primary_records"] == 2 * a.pairs
            if rep:
                ok = ok and chk["dup_flagged_primaries"] == 2 * rep["dups"] and chk["discordant_records"] == 2 * rep["discordant_pairs"] and chk["splitter_records"] == rep["splitter_lines"]
            out["invariants_ok"] = bool(ok)
            out["positions_nondecreasing_within_contig"] = srt.stdout.strip() == "1"
    print(json.dumps(out))
    td_obj.cleanup()


if __name__ == "__main__":
    main()
