#!/usr/bin/env python3
"""Randomised check of rank mode on the emulation build: the reference's script once (one pipeline) and through bin/speedseq-ranks (2-5 pipelines side
by side) on the same random input -- plain or gzip, interleaved or two files, read length, batch size, sort spills, scanner on / off drawn per run;
the three sorted BAMs must hold the same record bytes in the same order and the joined file's index must be what `sambamba index` makes of it.
usage: tools/fuzz_ranks.py [runs]"""
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import simreads  # noqa: E402
import test_ranks as T  # noqa: E402
from common import EXAMPLE_FA  # noqa: E402


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    bad = 0
    for seed in range(runs):
        rng = random.Random(1000 + seed)
        world = rng.choice([2, 2, 3, 3, 4, 5])
        n_pairs = rng.choice([300, 800, 1500, 3000])
        rl = rng.choice([100, 150, 150, 250])
        kw = {"read_len": rl} if rl != 150 else {}
        if rl == 250:
            kw.update(ins_mean=800, ins_std=150)
        gz, two = rng.random() < 0.5, rng.random() < 0.4
        chunk = str(rng.choice([15000, 40000, 90000, 250000]))
        extra = ""
        if rng.random() < 0.5:
            extra += "export SSG_SORT_CHUNK_BYTES=%d\n" % rng.choice([60000, 300000, 1500000])
        if rng.random() < 0.3:
            extra += "export SSG_RANKS_SPLIT=0\n"
        if rng.random() < 0.3:
            extra += "export SSG_FUSED_SHM=0\n"
        if rng.random() < 0.3:
            extra += "export SSG_BGZF_DEVICE=1\n"
        what = "seed %d: %d ranks, %d pairs 2x%d, %s%s, batches of %s bases x 2, %s" % (seed, world, n_pairs, rl, "gz" if gz else "plain", " two files" if two else " interleaved", chunk, extra.replace("export ", "").replace("\n", " "))
        with tempfile.TemporaryDirectory() as d:
            fq = os.path.join(d, "r_1.fq" + (".gz" if gz else "")); fq2 = os.path.join(d, "r_2.fq" + (".gz" if gz else "")) if two else None
            simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), n_pairs, seed=seed, **kw), interleaved=not two, path2=fq2)
            tail = ([] if two else ["-p"]) + ["-R", T.RG]
            outs = {}
            for tag, cmd0, cfg_extra in (("one", ["bash", T.REF_SCRIPT], extra.replace("export SSG_RANKS_SPLIT=0\n", "")), ("many", [os.path.join(ROOT, "bin", "speedseq-ranks"), "-n", str(world), "--script", T.REF_SCRIPT, "--"], extra)):
                cfg, ref, env = T._setup(os.path.join(d, tag), cfg_extra)
                env["SSG_BWA_CHUNK_BASES"] = chunk
                out = os.path.join(d, tag, "out")
                r = subprocess.run(cmd0 + ["align", "-K", cfg, "-o", out, "-M", "3", "-t", "2"] + tail + [ref, fq] + ([fq2] if two else []), cwd=os.path.join(d, tag), env=env, capture_output=True, text=True, timeout=900)
                if r.returncode != 0:
                    print("FAILED RUN", what, tag, r.stderr[-1500:]); bad += 1; break
                outs[tag] = out
            if len(outs) < 2:
                continue
            ok = all(T._records(outs["many"] + s) == T._records(outs["one"] + s) for s in (".bam", ".splitters.bam", ".discordants.bam"))
            mine = open(outs["many"] + ".bam.bai", "rb").read()
            subprocess.run([os.path.join(T.EMU, "sambamba_emu"), "index", outs["many"] + ".bam"], check=True)
            ok = ok and mine == open(outs["many"] + ".bam.bai", "rb").read()
            print(("ok   " if ok else "DIFF ") + what, flush=True)
            bad += 0 if ok else 1
    print("bad", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
