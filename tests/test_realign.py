"""the reference's own `speedseq realign` (bin/speedseq:1643-2030), run UNMODIFIED on this repo's executables:
bamkit's four helpers behind $PYTHON (bin/pyrun), bwa mem -C -p, samblaster, sambamba view / sort / index.  Upstream's helper
scripts (hall-lab/bamkit) are an empty submodule in the reference tree, so there is no reference output to pin bytes on; the checks
are the properties the path must have:
  * bamtofastq gives back exactly the reads that went into the BAM (original orientation, both mates, read group as comment),
  * realigning a BAM reproduces the alignment it came from (same insert-size batch, so the same decisions up to tie-breaks on the
    pair ordinal), with the read groups of the input BAM's header in the new header and on every record."""
import gzip
import os
import shutil
import subprocess

import pytest

import simreads
from common import EXAMPLE_FA, ROOT
import test_speedseq_script as ts

BAMKIT = os.path.join(ROOT, "bin", "bamkit")


def _need_bamkit():
    if not os.path.exists(BAMKIT):
        subprocess.check_call(["make", "-C", ROOT, "bin/bamkit"])


def _sam_to_bam(tmp, name, text, sambamba):
    sam = os.path.join(tmp, name + ".sam")
    open(sam, "w").write(text)
    bam = os.path.join(tmp, name + ".bam")
    subprocess.check_call("%s view -S -f bam %s > %s" % (sambamba, sam, bam), shell=True)
    return bam


HDR = "@HD\tVN:1.3\tSO:coordinate\n@SQ\tSN:c1\tLN:1000\n"
RG1 = "@RG\tID:a\tSM:s\tLB:L1\n@RG\tID:b\tSM:s\tLB:L2\n@RG\tID:c\tSM:s\tLB:L1\n"


def test_bamkit_header_tools(tmp_path, emu_lib):
    _need_bamkit()
    sambamba = os.path.join(ts.EMU, "sambamba_emu")
    t = str(tmp_path)
    b1 = _sam_to_bam(t, "x", HDR + RG1 + "@PG\tID:bwa\tPN:bwa\n", sambamba)
    b2 = _sam_to_bam(t, "y", HDR + "@RG\tID:d\tSM:s\tLB:L2\n@RG\tID:a\tSM:s\tLB:L1\n", sambamba)
    hdr = subprocess.check_output([BAMKIT, "bamcleanheader", b1, b2], text=True)
    assert hdr == HDR + RG1 + "@RG\tID:d\tSM:s\tLB:L2\n"           # first file's @HD/@SQ, distinct @RG lines, no @PG
    hp = os.path.join(t, "header.txt")
    open(hp, "w").write(hdr)
    assert subprocess.check_output([BAMKIT, "bamlibs", "-S", hp], text=True) == "a,c\nb,d\n"
    sam = "@SQ\tSN:c1\tLN:1000\n@PG\tID:bwa\tPN:bwa\nr1\t4\t*\t0\t0\t*\t*\t0\t0\tA\tI\tRG:Z:a\n"
    out = subprocess.run([BAMKIT, "bamheadrg", "-d", hp, "-r", "a,c"], input=sam, capture_output=True, text=True, check=True).stdout
    assert out == "@SQ\tSN:c1\tLN:1000\n@PG\tID:bwa\tPN:bwa\n@RG\tID:a\tSM:s\tLB:L1\n@RG\tID:c\tSM:s\tLB:L1\nr1\t4\t*\t0\t0\t*\t*\t0\t0\tA\tI\tRG:Z:a\n"
    out = subprocess.run([BAMKIT, "bamheadrg", "-d", hp], input="@SQ\tSN:c1\tLN:1000\n", capture_output=True, text=True, check=True).stdout
    assert out.count("@RG") == 4                                 # header-only stream, all read groups


def _fastq_records(path):
    recs = {}
    with gzip.open(path, "rt") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().strip(); f.readline(); q = f.readline().strip()
            name = h[1:].split()[0]
            end = 1
            if name.endswith("/1") or name.endswith("/2"):
                end = int(name[-1]); name = name[:-2]
            elif name in recs and 1 in recs[name]:
                end = 2
            recs.setdefault(name, {})[end] = (s, q)
    return recs


def test_bamtofastq_gives_back_the_reads(tmp_path, emu_lib):
    """align (reference script, emulated product) -> coordinate-sorted BAM -> bamtofastq: every pair comes back as it went in"""
    ts._need_tools(); _need_bamkit()
    fq = ts._fastq(tmp_path, 600)
    out = ts._run_align(str(tmp_path / "emu"), os.path.join(ts.EMU, "bwa_emu"), os.path.join(ts.EMU, "samblaster_emu"), fq, sambamba=os.path.join(ts.EMU, "sambamba_emu"))
    txt = subprocess.check_output([BAMKIT, "bamtofastq", "-r", "NA12878", out + ".bam"], text=True).split("\n")
    want = _fastq_records(fq)
    got = {}
    assert len(txt) % 8 == 1
    for i in range(0, len(txt) - 1, 8):
        h1, s1, _, q1, h2, s2, _, q2 = txt[i:i + 8]
        n1, c1 = h1[1:].split(" "); n2, c2 = h2[1:].split(" ")
        assert n1.endswith("/1") and n2.endswith("/2") and n1[:-2] == n2[:-2] and c1 == c2 == "RG:Z:NA12878"
        got[n1[:-2]] = {1: (s1, q1), 2: (s2, q2)}
    assert got == want
    renamed = subprocess.check_output([BAMKIT, "bamtofastq", "-n", out + ".bam"], text=True).split("\n")
    assert renamed[0].split(" ")[0] == "@0/1" and renamed[4].split(" ")[0] == "@0/2" and len(renamed) == len(txt)
    assert subprocess.check_output([BAMKIT, "bamtofastq", "-r", "other", out + ".bam"], text=True) == ""


def _run_realign(d, bam, bwa_cmd, samblaster_cmd, sambamba, n_threads=4):
    os.makedirs(d)
    bindir = os.path.join(d, "bin")
    os.makedirs(bindir)
    wrappers = [("bwa", bwa_cmd), ("samblaster", samblaster_cmd)] + [(t + ".py", BAMKIT + " " + t) for t in ("bamtofastq", "bamheadrg", "bamcleanheader", "bamlibs")]
    for name, cmd in wrappers:
        with open(os.path.join(bindir, name), "w") as f:
            f.write("#!/bin/sh\nexec %s \"$@\"\n" % cmd)
        os.chmod(os.path.join(bindir, name), 0o755)
    os.symlink(shutil.which("mawk"), os.path.join(bindir, "gawk"))
    cfg = os.path.join(d, "speedseq.config")
    with open(cfg, "w") as f:
        f.write("BWA=%s/bwa\nSAMBLASTER=%s/samblaster\nSAMBAMBA=%s\nPARALLEL=%s/bin/parallel\nPYTHON=%s/bin/pyrun\nMBUFFER=%s/bin/mbuffer\n" % (bindir, bindir, sambamba, ROOT, ROOT, ROOT))
        for t in ("BAMTOFASTQ", "BAMHEADRG", "BAMCLEANHEADER", "BAMLIBS"):
            f.write("%s=%s/%s.py\n" % (t, bindir, t.lower()))
    ref = os.path.join(d, "ref.fa")
    shutil.copy(EXAMPLE_FA, ref)
    env = dict(os.environ, PATH="%s:%s" % (bindir, os.environ["PATH"]))
    out = os.path.join(d, "again")
    r = subprocess.run(["bash", ts.REF_SCRIPT, "realign", "-K", cfg, "-o", out, "-M", "3", "-t", str(n_threads), ref, bam],
                       cwd=d, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return out


def _primary(bam):
    recs = {}
    for l in subprocess.check_output([ts.SAMTOOLS, "view", "-F", "0x900", bam], text=True).split("\n"):
        if l:
            f = l.split("\t")
            recs[(f[0], int(f[1]) & 0xc0)] = f
    return recs


def _check_realigned(out, src):
    for suffix in (".bam", ".splitters.bam", ".discordants.bam"):
        assert os.path.getsize(out + suffix) > 0 and os.path.exists(out + suffix + ".bai")
    hdr = subprocess.check_output([ts.SAMTOOLS, "view", "-H", out + ".bam"], text=True)
    assert "SO:coordinate" in hdr and "@RG\tID:NA12878\tSM:NA12878\tLB:lib1" in hdr          # from the input BAM's header, through bamheadrg
    a, b = _primary(src + ".bam"), _primary(out + ".bam")
    assert set(a) == set(b) and len(a) >= 1000
    same = sum(1 for k in a if a[k][2:4] == b[k][2:4] and a[k][5] == b[k][5] and (int(a[k][1]) & ~0x400) == (int(b[k][1]) & ~0x400))
    assert same >= 0.99 * len(a), (same, len(a))
    assert all("RG:Z:NA12878" in b[k][11:] for k in b)                                        # bwa mem -C: the FASTQ comment
    assert all(a[k][9] == b[k][9] and a[k][10] == b[k][10] for k in a)                        # the reads themselves
    nd_a = sum(1 for k in a if int(a[k][1]) & 0x400); nd_b = sum(1 for k in b if int(b[k][1]) & 0x400)
    assert abs(nd_a - nd_b) <= 0.05 * max(nd_a, 1) + 2, (nd_a, nd_b)                          # which copy of a duplicate set is kept follows the new order


def test_reference_realign_script_with_product_sources_emulated(tmp_path, emu_lib):
    ts._need_tools(); _need_bamkit()
    fq = ts._fastq(tmp_path, 800)
    emu = lambda n: os.path.join(ts.EMU, n)
    src = ts._run_align(str(tmp_path / "align"), emu("bwa_emu"), emu("samblaster_emu"), fq, sambamba=emu("sambamba_emu"))
    out = _run_realign(str(tmp_path / "realign"), src + ".bam", emu("bwa_emu"), emu("samblaster_emu"), emu("sambamba_emu"))
    _check_realigned(out, src)


@pytest.mark.gpu
def test_reference_realign_script_with_product_executables(tmp_path, gpu_lib):
    ts._need_tools(); _need_bamkit()
    fq = ts._fastq(tmp_path, 3000)
    b = lambda n: os.path.join(ROOT, "bin", n)
    src = ts._run_align(str(tmp_path / "align"), b("bwa"), b("samblaster"), fq, sambamba=b("sambamba"))
    out = _run_realign(str(tmp_path / "realign"), src + ".bam", b("bwa"), b("samblaster"), b("sambamba"))
    _check_realigned(out, src)


def test_reference_realign_script_two_libraries_emulated(tmp_path, emu_lib):
    """a BAM with two read groups in two libraries: the script realigns each library on its own (bamlibs, bamtofastq -r), then
    `sambamba merge`s the three kinds of BAM and indexes them (bin/speedseq:1986-2025)"""
    ts._need_tools(); _need_bamkit()
    emu = lambda n: os.path.join(ts.EMU, n)
    srcs = []
    for k, (rgid, lib, seed) in enumerate((("rgA", "L1", 41), ("rgB", "L2", 42))):
        fq = str(tmp_path / ("reads%d.fq.gz" % k))
        simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 400, seed=seed, prefix="q%d_" % k))
        srcs.append(ts._run_align(str(tmp_path / ("align%d" % k)), emu("bwa_emu"), emu("samblaster_emu"), fq, sambamba=emu("sambamba_emu"),
                                  rg="@RG\\tID:%s\\tSM:s\\tLB:%s" % (rgid, lib)))
    merged = str(tmp_path / "both.bam")
    subprocess.check_call([emu("sambamba_emu"), "merge", "-t", "2", merged, srcs[0] + ".bam", srcs[1] + ".bam"])
    out = _run_realign(str(tmp_path / "realign"), merged, emu("bwa_emu"), emu("samblaster_emu"), emu("sambamba_emu"))
    for suffix in (".bam", ".splitters.bam", ".discordants.bam"):
        assert os.path.getsize(out + suffix) > 0 and os.path.exists(out + suffix + ".bai")
    hdr = subprocess.check_output([ts.SAMTOOLS, "view", "-H", out + ".bam"], text=True)
    assert "@RG\tID:rgA\tSM:s\tLB:L1" in hdr and "@RG\tID:rgB\tSM:s\tLB:L2" in hdr and "SO:coordinate" in hdr
    a, b = _primary(merged), _primary(out + ".bam")
    assert set(a) == set(b) and len(a) == 1600
    rg = lambda f: [t for t in f[11:] if t.startswith("RG:Z:")][0]
    assert all(rg(a[k]) == rg(b[k]) for k in a)
    assert all(rg(b[k]) == ("RG:Z:rgA" if k[0].startswith("q0_") else "RG:Z:rgB") for k in b)
    same = sum(1 for k in a if a[k][2:4] == b[k][2:4] and a[k][5] == b[k][5])
    assert same >= 0.99 * len(a), (same, len(a))
    pos = [(l.split("\t")[2], int(l.split("\t")[3])) for l in subprocess.check_output([ts.SAMTOOLS, "view", "-F", "4", out + ".bam"], text=True).split("\n") if l]
    assert pos == sorted(pos, key=lambda x: (x[0] != "20_slice", x[1]))                      # merged output is coordinate-sorted
