#!/usr/bin/env python3
"""Randomised end-to-end check of the host pipeline on the emulation build: `bwa mem | samblaster | sambamba view | sambamba sort` with the
text hand-off against the fused one, every switch of the host side drawn at random per run (emulated devices, calls in flight, formatter
threads, frame segments on / off, gzip decoding threads, parse threads and their slab / piece sizes, the moment of the suffix-array
densification, sort spills and their genome ranges, samblaster's options).  The sorted BAM's record bytes and both side streams must be
equal.  usage: tools/fuzz_pipeline.py [runs [BINDIR]]   (BINDIR: the product's executables bwa / samblaster / sambamba, e.g. bin/ on a GPU box; default the emulation build)"""
import gzip
import os
import random
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import simreads  # noqa: E402
from common import EXAMPLE_FA  # noqa: E402

EMU = os.path.join(ROOT, "tests", "emu")


def recs(bam):
    raw = gzip.open(bam, "rb").read()
    l_text, = struct.unpack_from("<i", raw, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, o)
    o += 4
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", raw, o)
        o += 4 + l + 4
    return raw[o:]


def main(runs=None, bindir=None, first_seed=0):
    runs = 8 if runs is None else runs
    exe = (lambda n: os.path.join(bindir, n)) if bindir else (lambda n: os.path.join(EMU, n + "_emu"))
    bad = 0
    for seed in range(first_seed, first_seed + runs):
        rng = random.Random(seed)
        with tempfile.TemporaryDirectory() as d:
            fq = os.path.join(d, "r.fq.gz")
            n = rng.randint(200, 900)
            simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), n, seed=100 + seed, chim_frac=rng.random() * 0.1, disc_frac=rng.random() * 0.1, dup_frac=rng.random() * 0.3))
            opts = rng.choice([[], ["--addMateTags"], ["--excludeDups", "--addMateTags", "--maxSplitCount", "2", "--minNonOverlap", "20"]])
            chunk_bases = str(rng.choice([8000, 30000, 100000]))   # upstream's batch size: the scope of the insert-size model, i.e. part of the INPUT -- the same for both runs
            outs = {}
            for mode in ("text", "fused"):
                env = dict(os.environ, SSG_BWA_CHUNK_BASES=chunk_bases, SSG_BWA_CALL_PAIRS=str(rng.choice([50, 200, 1000])),
                           SSG_EMU_DEVICES=str(rng.choice([1, 2, 3])), SSG_BWA_INFLIGHT=str(rng.choice([1, 2])), SSG_BWA_FORMATTERS=str(rng.choice([1, 3])),
                           SSG_FUSED_SHM_MIN=str(rng.choice([1, 10 ** 9])), SSG_GZ_THREADS=str(rng.choice([1, 3])), SSG_GZ_CHUNK="30000",
                           SSG_FASTQ_THREADS=str(rng.choice([1, 3])), SSG_FASTQ_SLAB="40000", SSG_FASTQ_PIECE="5000", SSG_BWA_DENSIFY_AFTER=str(rng.choice([0, 100, 10 ** 9])))
                if mode == "fused":
                    env["SSG_FUSED"] = "1"
                spl, disc = os.path.join(d, mode + ".spl"), os.path.join(d, mode + ".disc")
                p1 = subprocess.run([exe("bwa"), "mem", "-t", "2", "-p", "-R", "@RG\\tID:g\\tSM:s\\tLB:l", EXAMPLE_FA, fq], capture_output=True, env=env, timeout=300)
                p2 = subprocess.run([exe("samblaster")] + opts + ["--splitterFile", spl, "--discordantFile", disc], input=p1.stdout, capture_output=True, env=env, timeout=300)
                p3 = subprocess.run([exe("sambamba"), "view", "-S", "-f", "bam", "-l", "0", "/dev/stdin"], input=p2.stdout, capture_output=True, env=env, timeout=300)
                env2 = dict(env, SSG_SORT_CHUNK_BYTES=str(rng.choice([100000, 10 ** 9])), SSG_SORT_RANGES=str(rng.choice([1, 5, 40])))
                p4 = subprocess.run([exe("sambamba"), "sort", "-t", "3", "-m", "1G", "--tmpdir", os.path.join(d, mode + "tmp"), "-o", os.path.join(d, mode + ".bam"), "/dev/stdin"],
                                    input=p3.stdout, capture_output=True, env=env2, timeout=300)
                if any(p.returncode for p in (p1, p2, p3, p4)):
                    bad += 1
                    print(seed, mode, "rc", [p.returncode for p in (p1, p2, p3, p4)], p1.stderr[-200:], p4.stderr[-200:])
                    break
                strip = lambda x: "\n".join(l for l in x.split("\n") if not l.startswith("@PG"))  # noqa: E731
                outs[mode] = (recs(os.path.join(d, mode + ".bam")), strip(open(spl).read()), strip(open(disc).read()))
            if len(outs) == 2:
                if outs["text"] != outs["fused"]:
                    bad += 1
                    print(seed, "DIFF", [a == b for a, b in zip(outs["text"], outs["fused"])])
                else:
                    print(seed, "ok", n, "pairs,", len(outs["text"][0]), "bytes of records", flush=True)
    print("bad", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else None, sys.argv[2] if len(sys.argv) > 2 else None))
