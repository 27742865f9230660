#!/usr/bin/env python3
"""Host-side soak of the fused pipeline's stages behind `bwa mem` (no GPU needed: the emulation build decides and sorts): the BATCH frames
of 100 k simulated pairs, repeated `--reps` x 10 times (a million pairs per 10), through samblaster | sambamba view | sambamba sort with
the given -m.  What it is for: more than 4 GiB of records and tens of millions of them through the record store, the spill-and-merge
path and the in-memory path, frame payloads as mapped segments; checks the record count, coordinate order (the reference's samtools)
and that the .bai written by the sort equals the one `sambamba index` makes.  The alignment itself is covered elsewhere; every copy
after the first is a duplicate, which exercises the duplicate table, not the aligner.
usage: tools/soak_host.py [--reps 16] [--mem 4G] [--keep]"""
import argparse
import os
import shutil
import struct
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
EMU = os.path.join(ROOT, "tests", "emu")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=16, help="millions of pairs")
    ap.add_argument("--mem", default="4G", help="-m of sambamba sort (small: spills; large: one in-memory run)")
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    import simreads
    from common import EXAMPLE_FA
    td = tempfile.mkdtemp(prefix="ssg_soak_", dir="/tmp")
    f1, f2 = os.path.join(td, "r1.fq"), os.path.join(td, "r2.fq")
    simreads.write_fastq(f1, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 100000, seed=5), interleaved=False, path2=f2)
    t0 = time.time()
    base = subprocess.run([os.path.join(EMU, "bwa_emu"), "mem", "-t", "8", "-R", "@RG\\tID:g\\tSM:s\\tLB:l", EXAMPLE_FA, f1, f2],
                          env=dict(os.environ, SSG_FUSED="1", SSG_FUSED_SHM="0"), capture_output=True, check=True).stdout
    print("100 k pairs aligned by the emulation build in %.0f s: %d bytes of frames" % (time.time() - t0, len(base)), flush=True)
    o, frames = 8, []
    while o < len(base):
        t, z, l = struct.unpack_from("<IIQ", base, o)
        frames.append((t, base[o:o + 16 + l]))
        o += 16 + l
    batches = [f for t, f in frames if t == 2]
    n_rec = sum(struct.unpack_from("<Q", f, 16)[0] for f in batches)
    out = os.path.join(td, "out.bam")
    env = dict(os.environ, SSG_FUSED="1", SSG_SORT_LOG="1", SSG_SBL_LOG="1")
    p1 = subprocess.Popen([os.path.join(EMU, "samblaster_emu"), "--excludeDups", "--addMateTags", "--maxSplitCount", "2", "--minNonOverlap", "20",
                           "--splitterFile", os.path.join(td, "spl.sam"), "--discordantFile", os.path.join(td, "disc.sam")], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
    p2 = subprocess.Popen([os.path.join(EMU, "sambamba_emu"), "view", "-S", "-f", "bam", "-l", "0", "/dev/stdin"], stdin=p1.stdout, stdout=subprocess.PIPE, env=env)
    p3 = subprocess.Popen([os.path.join(EMU, "sambamba_emu"), "sort", "-t", str(os.cpu_count() or 8), "-m", a.mem, "--tmpdir", os.path.join(td, "tmp"), "-o", out, "/dev/stdin"], stdin=p2.stdout, env=env)
    p1.stdout.close(); p2.stdout.close()
    t0 = time.time()
    w = p1.stdin
    w.write(base[:8]); w.write(frames[0][1])
    for _ in range(10 * a.reps):
        for f in batches:
            w.write(f)
    w.write(frames[-1][1]); w.close()
    rcs = [p.wait() for p in (p1, p2, p3)]
    wall = time.time() - t0
    want = n_rec * 10 * a.reps
    print("pipeline rc %s, %.1f s for %d M pairs (%.1f GB of records)" % (rcs, wall, a.reps, sum(len(f) for f in batches) * 10 * a.reps / 1e9), flush=True)
    assert rcs == [0, 0, 0]
    samtools = os.path.join(ROOT, "oracle", "_ref", "samtools")
    chk = subprocess.run("%s view %s | awk 'BEGIN{ok=1} { if ($3==pc && $4<pp) ok=0; pc=$3; pp=$4 } END{print ok, NR}'" % (samtools, out), shell=True, capture_output=True, text=True, check=True).stdout.split()
    print("records %s (expected %d), coordinate order %s, %d bytes of BAM" % (chk[1], want, "ok" if chk[0] == "1" else "BROKEN", os.path.getsize(out)))
    assert chk == ["1", str(want)]
    had_bai = os.path.exists(out + ".bai")
    if had_bai:
        shutil.copy(out + ".bai", out + ".bai.sort")
        for x in (".bai", ".bai.ssg"):
            if os.path.exists(out + x):
                os.remove(out + x)
    subprocess.run([os.path.join(EMU, "sambamba_emu"), "index", out], check=True)
    if had_bai:
        assert open(out + ".bai", "rb").read() == open(out + ".bai.sort", "rb").read()
        print(".bai written by the sort == .bai of `sambamba index`")
    else:
        print("spill-and-merge run: the index comes from `sambamba index` (%d bytes)" % os.path.getsize(out + ".bai"))
    left = [f for f in os.listdir(os.environ.get("SSG_FUSED_SHM", "/dev/shm")) if f.startswith("ssgfuse.")] if os.path.isdir(os.environ.get("SSG_FUSED_SHM", "/dev/shm")) else []
    print("segments left behind: %d" % len(left))
    if not a.keep:
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    main()
