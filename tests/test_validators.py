"""Parity at the bench's shape + oracle-independent validation: a many-contig reference with
ambiguity holes and reads that straddle contig ends, through the executables (`bwa index`, `bwa mem | samblaster`).
CPU side: host-emulation build on a small genome; -m gpu: 27 contigs / 120 Mbp on the MI355X.  The product's three SAM
streams must equal the oracle's, and tests/validators.py re-derives MD / NM / AS / mate fields / positions / duplicates
from the reference bases and the simulator's truth without the oracle."""
import os
import subprocess

import numpy as np
import pytest

import simreads
import validators
from common import ROOT

ORC = os.path.join(ROOT, "oracle", "orc_bwa")
EMU = os.path.join(ROOT, "tests", "emu")
COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def make_reference(path, n_contigs, total, seed):
    """contigs of uneven length, 'N' runs inside and at the ends (as centromere / telomere gaps), a planted repeat family"""
    rng = np.random.default_rng(seed)
    w = rng.random(n_contigs) + 0.3
    lens = np.maximum(1500, (w / w.sum() * total).astype(np.int64))
    fam = rng.integers(0, 4, size=400).astype(np.uint8)
    contigs = []
    for i, L in enumerate(lens):
        s = rng.integers(0, 4, size=int(L)).astype(np.uint8)
        for _ in range(max(1, int(L) // 40000)):
            p = int(rng.integers(0, L - 400)); c = fam.copy(); m = rng.random(400) < 0.03; c[m] = rng.integers(0, 4, size=int(m.sum())); s[p:p + 400] = c
        if i % 3 == 0:
            s[:int(rng.integers(5, 60))] = 4                 # leading gap
        if i % 4 == 1:
            s[-int(rng.integers(5, 60)):] = 4                # trailing gap
        for _ in range(1 + int(L) // 200000):
            p = int(rng.integers(200, L - 200)); s[p:p + int(rng.integers(1, 120))] = 4
        contigs.append(("ctg%d" % (i + 1) if i % 5 else "GL%d.1" % (i + 1), s))
    simreads.write_fasta(path, contigs)
    return contigs


def boundary_pairs(contigs, rl, seed):
    """read 1 runs over the end of contig i into the start of contig i+1 (adjacent in the packed reference); read 2 lies inside contig i"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(len(contigs) - 1):
        a, b = contigs[i][1], contigs[i + 1][1]
        k = int(rng.integers(30, rl - 30))
        r1 = np.concatenate([a[-k:], b[:rl - k]])
        p2 = max(0, a.size - 420)
        r2 = COMP[a[p2:p2 + rl][::-1]]
        out.append(("b%d_%s_%d_%d_b" % (i, contigs[i][0], a.size - k + 1, a.size), r1.copy(), r2.copy()))
    return out


def _pipeline(bwa, sbl, ref, fq, d, tag, threads=4):
    o, sp, di = (os.path.join(d, tag + x) for x in (".sam", ".spl.sam", ".disc.sam"))
    cmd = "%s mem -t %d -p -R '@RG\\tID:x\\tSM:y' %s %s 2>/dev/null | %s --excludeDups --addMateTags --maxSplitCount 2 --minNonOverlap 20 --splitterFile %s --discordantFile %s > %s 2>/dev/null" % (
        bwa, threads, ref, fq, sbl, sp, di, o)
    subprocess.check_call(["bash", "-c", "set -o pipefail; " + cmd])
    return [[l for l in open(p).read().split("\n") if not l.startswith("@PG")] for p in (o, sp, di)]


def _run(tmp_path, bwa, sbl, n_contigs, total, n_pairs, compare_index):
    d = str(tmp_path)
    ref = os.path.join(d, "ref.fa")
    contigs = make_reference(ref, n_contigs, total, seed=9)
    subprocess.check_call([bwa, "index", ref], stderr=subprocess.DEVNULL)         # the product's `bwa index`: holes -> .amb, random bases in .pac
    if compare_index:
        oref = os.path.join(d, "oref.fa")
        os.symlink(ref, oref)
        subprocess.check_call([ORC, "index", oref], stderr=subprocess.DEVNULL)
        for ext in ("amb", "ann", "bwt", "pac", "sa"):
            assert open(ref + "." + ext, "rb").read() == open(oref + "." + ext, "rb").read(), ext
    assert int(open(ref + ".amb").read().split()[2]) > n_contigs // 2                   # holes were recorded
    pairs = simreads.simulate(contigs, n_pairs, seed=10, chim_frac=0.02, disc_frac=0.02) + boundary_pairs(contigs, 150, seed=11)
    fq = os.path.join(d, "reads.fq")
    simreads.write_fastq(fq, pairs)
    got = _pipeline(bwa, sbl, ref, fq, d, "got")
    exp = _pipeline(ORC, ORC + " samblaster", ref, fq, d, "exp")
    for g, e, what in zip(got, exp, ("samblaster stdout", "splitters", "discordants")):
        assert g == e, what
    # ---- oracle-independent validation of the product's output ----
    text = "\n".join(got[0])
    pac = validators.pac_contigs(ref)
    n, bad = validators.md_nm_consistency(text, pac)
    assert n > n_pairs and bad == 0, ("MD/NM", n, bad)
    n, bad = validators.as_from_cigar(text, pac)
    assert n > n_pairs and bad <= n // 1000, ("AS", n, bad)
    n, bad = validators.mate_symmetry(text)
    assert n > 2 * n_pairs and bad == 0, ("mate fields", n, bad)
    n, hit = validators.truth_recall(text)
    assert n > n_pairs and hit >= 0.99 * n, ("recall", n, hit)
    marked, planted_marked, missed = validators.planted_duplicates(text)
    assert marked > n_pairs // 50 and planted_marked >= 0.98 * marked and missed <= max(2, marked // 50), ("dups", marked, planted_marked, missed)
    assert len(got[1]) > len(contigs) and len(got[2]) > len(contigs)                   # side streams are populated (chimeric / discordant / boundary reads)
    return text


def test_many_contigs_holes_boundaries_emu(tmp_path, emu_lib):
    _run(tmp_path, os.path.join(EMU, "bwa_emu"), os.path.join(EMU, "samblaster_emu"), 27, 160000, 700, compare_index=True)


@pytest.mark.gpu
def test_many_contigs_holes_boundaries_gpu(tmp_path, gpu_lib):
    _run(tmp_path, os.path.join(ROOT, "bin", "bwa"), os.path.join(ROOT, "bin", "samblaster"), 27, 120000000, 30000, compare_index=False)
