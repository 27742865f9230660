"""The pin that closes the day upstream's sources (or binaries) are at hand.

`/root/reference/src/bwa` and `src/samblaster` are empty submodules (`.gitmodules:4-6,16-18` name lh3/bwa and GregoryFaust/samblaster), so
the oracle is a restatement checked against reference-held bytes only where the reference holds some (index files, BAM, kseq / ksort).
These tests run the SAME parity checks the suite runs against the oracle against the REAL executables instead:

    SSG_UPSTREAM_BWA=/path/to/bwa  SSG_UPSTREAM_SAMBLASTER=/path/to/samblaster  python -m pytest tests/test_swap_in_upstream.py

  * upstream `bwa mem` vs the oracle's command line (pins oracle/orc_*.c: every "0 differing" of the suite then is a statement about
    upstream) and vs the product's `bwa` (host emulation here, bin/bwa under `-m gpu`);
  * upstream `bwa index` vs the bundled example index and vs the device index builder;
  * upstream `samblaster` on the product's SAM vs the oracle's and the product's samblaster: stdout, splitters, discordants.
Without the variables every test here is skipped (and says which variable it wants).  bwa 0.7.12 is the version the restatement follows
(oracle/README.md lists the thirteen places where versions differ and the choice made at each)."""
import os
import subprocess

import pytest

import simreads
from common import EXAMPLE_FA, ROOT

UP_BWA = os.environ.get("SSG_UPSTREAM_BWA")
UP_SBL = os.environ.get("SSG_UPSTREAM_SAMBLASTER")
ORC = os.path.join(ROOT, "oracle", "orc_bwa")
EMU = os.path.join(ROOT, "tests", "emu")
RG = "@RG\\tID:grp1\\tSM:s1\\tLB:lib1"
SBL_ARGS = ["--excludeDups", "--addMateTags", "--maxSplitCount", "2", "--minNonOverlap", "20"]   # reference bin/speedseq:439

need_bwa = pytest.mark.skipif(not UP_BWA, reason="set SSG_UPSTREAM_BWA to an lh3/bwa executable (0.7.12) to pin the oracle and the product against upstream")
need_sbl = pytest.mark.skipif(not UP_SBL, reason="set SSG_UPSTREAM_SAMBLASTER to a GregoryFaust/samblaster executable to pin the duplicate / discordant / splitter marking against upstream")


def _no_pg(text):
    return "\n".join(l for l in text.split("\n") if not l.startswith("@PG"))


def _reads(tmp_path, n_pairs, seed, **kw):
    fq = os.path.join(str(tmp_path), "reads_%d.fq" % seed)
    simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), n_pairs, seed=seed, **kw))
    return fq


def _mem(exe, fq, extra=()):
    return _no_pg(subprocess.run(list(exe) + ["mem", "-t", "4", "-p", "-R", RG] + list(extra) + [EXAMPLE_FA, fq], capture_output=True, check=True).stdout.decode())


def _sbl(exe, sam, d, tag):
    spl, disc = os.path.join(d, tag + ".spl.sam"), os.path.join(d, tag + ".disc.sam")
    out = subprocess.run(list(exe) + SBL_ARGS + ["--splitterFile", spl, "--discordantFile", disc], input=sam.encode(), capture_output=True, check=True).stdout.decode()
    return _no_pg(out), _no_pg(open(spl).read()), _no_pg(open(disc).read())


@need_bwa
@pytest.mark.parametrize("seed,kw,extra", [(901, {}, ()), (902, {"read_len": 250, "ins_mean": 800, "ins_std": 150}, ()), (903, {}, ("-I", "400,50"))])
def test_upstream_bwa_mem_pins_the_oracle_and_the_emulated_product(tmp_path, emu_lib, seed, kw, extra):
    fq = _reads(tmp_path, 2000, seed, **kw)
    up = _mem([UP_BWA], fq, extra)
    assert up.count("\n") > 4000
    assert _mem([ORC], fq, extra) == up, "oracle/orc_bwa differs from upstream bwa mem"
    assert _mem([os.path.join(EMU, "bwa_emu")], fq, extra) == up, "the product (host emulation) differs from upstream bwa mem"


@need_bwa
@pytest.mark.parametrize("opts", [("-M",), ("-Y",), ("-S",), ("-P",), ("-A", "2"), ("-A", "2", "-B", "5", "-O", "7,9", "-E", "2,1"), ("-k", "25", "-c", "50", "-D", "0.3"), ("-L", "3,8", "-U", "9"), ("-T", "45", "-h", "2")],
                         ids=lambda o: "".join(o))
def test_upstream_bwa_mem_option_letters_pin_the_oracle(tmp_path, opts):
    """the option letters restated late in round 4 (oracle/README.md row 10c): upstream's own reading of them"""
    fq = _reads(tmp_path, 1500, 907)
    assert _mem([ORC], fq, opts) == _mem([UP_BWA], fq, opts), "oracle/orc_bwa differs from upstream bwa mem under " + " ".join(opts)


@need_bwa
def test_upstream_bwa_mem_single_end_pins_the_oracle(tmp_path):
    """one FASTQ without -p (oracle/README.md row 10d)"""
    fq = _reads(tmp_path, 1500, 908)
    run = lambda exe: _no_pg(subprocess.run(list(exe) + ["mem", "-t", "4", "-R", RG, EXAMPLE_FA, fq], capture_output=True, check=True).stdout.decode())
    assert run([ORC]) == run([UP_BWA])


@need_bwa
@pytest.mark.gpu
def test_upstream_bwa_mem_pins_the_product_on_the_gpu(tmp_path, gpu_lib):
    fq = _reads(tmp_path, 20000, 904)
    assert _mem([os.path.join(ROOT, "bin", "bwa")], fq) == _mem([UP_BWA], fq)


@need_bwa
def test_upstream_bwa_index_bytes(tmp_path, emu_lib):
    """upstream `bwa index` of the bundled FASTA: the five files byte for byte against the bundled ones (the reference's own) and the device builder's"""
    import shutil
    fa = os.path.join(str(tmp_path), "ref.fa")
    shutil.copy(EXAMPLE_FA, fa)
    subprocess.run([UP_BWA, "index", fa], check=True, capture_output=True)
    mine = os.path.join(str(tmp_path), "mine.fa")
    shutil.copy(EXAMPLE_FA, mine)
    subprocess.run([os.path.join(EMU, "bwa_emu"), "index", mine], check=True, capture_output=True)
    for ext in (".bwt", ".sa", ".pac", ".ann", ".amb"):
        up = open(fa + ext, "rb").read()
        assert up == open(EXAMPLE_FA + ext, "rb").read(), "bundled example index differs from upstream's " + ext
        assert up == open(mine + ext, "rb").read(), "device index builder differs from upstream's " + ext


@need_sbl
def test_upstream_samblaster_pins_the_oracle_and_the_emulated_product(tmp_path, emu_lib):
    d = str(tmp_path)
    fq = _reads(tmp_path, 3000, 905, dup_frac=0.08) if "dup_frac" in simreads.simulate.__code__.co_varnames else _reads(tmp_path, 3000, 905)
    sam = subprocess.run([ORC, "mem", "-t", "4", "-p", "-R", RG, EXAMPLE_FA, fq], capture_output=True, check=True).stdout.decode()
    up = _sbl([UP_SBL], sam, d, "up")
    assert up[0].count("\n") > 5000
    for exe, what in (([ORC, "samblaster"], "oracle samblaster"), ([os.path.join(EMU, "samblaster_emu")], "the product's samblaster (host emulation)")):
        got = _sbl(exe, sam, d, "got")
        for g, u, stream in zip(got, up, ("stdout", "splitters", "discordants")):
            assert g == u, "%s differs from upstream samblaster on %s" % (what, stream)


@need_sbl
@pytest.mark.gpu
def test_upstream_samblaster_pins_the_product_on_the_gpu(tmp_path, gpu_lib):
    d = str(tmp_path)
    fq = _reads(tmp_path, 20000, 906)
    sam = subprocess.run([os.path.join(ROOT, "bin", "bwa"), "mem", "-t", "4", "-p", "-R", RG, EXAMPLE_FA, fq], capture_output=True, check=True).stdout.decode()
    up, got = _sbl([UP_SBL], sam, d, "up"), _sbl([os.path.join(ROOT, "bin", "samblaster")], sam, d, "got")
    for g, u, stream in zip(got, up, ("stdout", "splitters", "discordants")):
        assert g == u, stream
