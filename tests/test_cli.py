"""Process-level drop-in boundary: the `bwa` and `samblaster` executables built on
libssgpu must emit the same three SAM streams as the CPU oracle's command line, byte for byte
(modulo the @PG lines).  CPU-side: the host-emulation build; `-m gpu`: the HIP build in bin/."""
import os
import subprocess

import pytest

import simreads
from common import EXAMPLE_FA, ROOT

ORC = os.path.join(ROOT, "oracle", "orc_bwa")
RG = "@RG\\tID:grp1\\tSM:s1\\tLB:lib1"


def _no_pg(text):
    return "\n".join(l for l in text.split("\n") if not l.startswith("@PG"))


def _run_pipeline(bwa, samblaster, ref, fq, d, tag, extra_bwa=()):
    spl, disc = os.path.join(d, tag + ".spl.sam"), os.path.join(d, tag + ".disc.sam")
    p1 = subprocess.run(bwa + ["mem", "-t", "4", "-p", "-R", RG] + list(extra_bwa) + [ref, fq], capture_output=True, check=True)
    p2 = subprocess.run(samblaster + ["--excludeDups", "--addMateTags", "--maxSplitCount", "2", "--minNonOverlap", "20",
                                      "--splitterFile", spl, "--discordantFile", disc], input=p1.stdout, capture_output=True, check=True)
    return p1.stdout.decode(), p2.stdout.decode(), open(spl).read(), open(disc).read()


def _check(bwa, samblaster, tmp_path, n_pairs, seed, extra_bwa=()):
    d = str(tmp_path)
    contigs = simreads.read_fasta(EXAMPLE_FA)
    fq = os.path.join(d, "reads.fq.gz")
    simreads.write_fastq(fq, simreads.simulate(contigs, n_pairs, seed=seed))
    got = _run_pipeline(bwa, samblaster, EXAMPLE_FA, fq, d, "got", extra_bwa)
    exp = _run_pipeline([ORC], [ORC, "samblaster"], EXAMPLE_FA, fq, d, "exp", extra_bwa)
    for g, e, what in zip(got, exp, ("bwa mem", "samblaster stdout", "splitters", "discordants")):
        assert _no_pg(g) == _no_pg(e), what
    assert "\t1" in got[1]   # sanity: SAM with flags


def test_cli_emu_matches_oracle(tmp_path, emu_lib):
    emu = os.path.join(ROOT, "tests", "emu")
    _check([os.path.join(emu, "bwa_emu")], [os.path.join(emu, "samblaster_emu")], tmp_path, 300, seed=21)


def test_cli_emu_samblaster_small_chunks(tmp_path, emu_lib, monkeypatch):
    """device calls of 7 blocks each: the duplicate set in HBM must carry first-seen-wins across calls (and grow)"""
    monkeypatch.setenv("SSG_SBL_CHUNK", "7")
    monkeypatch.setenv("SSG_SBL_TABLE_SLOTS", "16")
    emu = os.path.join(ROOT, "tests", "emu")
    _check([os.path.join(emu, "bwa_emu")], [os.path.join(emu, "samblaster_emu")], tmp_path, 200, seed=24)


def test_cli_emu_insert_override(tmp_path, emu_lib):
    emu = os.path.join(ROOT, "tests", "emu")
    _check([os.path.join(emu, "bwa_emu")], [os.path.join(emu, "samblaster_emu")], tmp_path, 120, seed=22, extra_bwa=("-I", "400,50"))


def test_cli_error_behaviour(tmp_path, emu_lib):
    emu = os.path.join(ROOT, "tests", "emu", "bwa_emu")
    r = subprocess.run([emu, "mem", "-R", "RG\\tID:x", EXAMPLE_FA, "/dev/null"], capture_output=True)
    assert r.returncode != 0 and b"not started with @RG" in r.stderr
    r = subprocess.run([emu, "mem", "-p", EXAMPLE_FA + ".missing", "/dev/null"], capture_output=True)
    assert r.returncode != 0


@pytest.mark.gpu
def test_cli_gpu_matches_oracle(tmp_path, gpu_lib):
    _check([os.path.join(ROOT, "bin", "bwa")], [os.path.join(ROOT, "bin", "samblaster")], tmp_path, 6000, seed=23)
