"""MI355X twins of tests that so far ran on the host emulation only ("what's weak" 4): samblaster's option sets, the
FASTQ grammar and two-file input through bin/bwa, the fused hand-off under other samblaster options and with every frame forced into a
mapped segment, the several-thread FASTQ reader.  Collected last: they were written after the round's last GPU run, so nothing earlier
in `pytest -m gpu -x` waits on them."""
import os
import subprocess

import pytest

import simreads
import test_cli
import test_fused
from common import EXAMPLE_FA, ROOT

DRY = bool(os.environ.get("SSG_TWINS_EMU"))   # dry run of this file's own code on the host emulation: SSG_TWINS_EMU=1 pytest tests/test_zz_twins_gpu.py -m gpu
B = (lambda n: os.path.join(ROOT, "tests", "emu", n + "_emu")) if DRY else (lambda n: os.path.join(ROOT, "bin", n))   # noqa: E731
pytestmark = pytest.mark.gpu


@pytest.fixture
def gpu_lib(request):
    return request.getfixturevalue("emu_lib") if DRY else request.getfixturevalue("gpu_lib_session")


@pytest.mark.parametrize("opts", [[], ["--addMateTags"], ["--excludeDups", "--addMateTags", "--maxSplitCount", "1", "--minNonOverlap", "50"],
                                  ["--excludeDups", "--maxSplitCount", "3", "--minNonOverlap", "5"]])
def test_gpu_samblaster_option_sets(tmp_path, gpu_lib, opts):
    test_cli._option_sets(tmp_path, B("samblaster"), opts)


@pytest.mark.parametrize("style", ["wrapped", "fasta", "crlf", "mixed"])
def test_gpu_fastq_grammar(tmp_path, gpu_lib, style):
    test_cli._grammar(tmp_path, B("bwa"), style)


def test_gpu_two_files_and_truncation(tmp_path, gpu_lib):
    test_cli._two_files(tmp_path, B("bwa"))


@pytest.mark.parametrize("opts", [[], ["--excludeDups", "--addMateTags", "--maxSplitCount", "1", "--minNonOverlap", "50"]])
def test_gpu_fused_samblaster_option_sets(tmp_path, gpu_lib, opts):
    test_fused._fused_option_sets(tmp_path, opts, B("bwa"), B("samblaster"), B("sambamba"), n_pairs=3000)


@pytest.mark.parametrize("seg", ["segments", "pipe_only"])
def test_gpu_fused_frames_as_mapped_segments(tmp_path, gpu_lib, seg):
    test_fused.T._need_tools()
    test_fused._fused_segments(tmp_path, seg, B("bwa"), B("samblaster"), B("sambamba"), n_pairs=3000)


def test_gpu_bwa_same_sam_with_fastq_pieces(tmp_path, gpu_lib):
    """bin/bwa on two plain files: the SAM text does not depend on how the several-thread reader cut them"""
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    simreads.write_fastq(f1, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 3000, seed=77), interleaved=False, path2=f2)
    outs = []
    for threads, piece in ((1, None), (4, 30000)):
        env = dict(os.environ, SSG_FASTQ_THREADS=str(threads))
        if piece:
            env["SSG_FASTQ_PIECE"] = str(piece)
        r = subprocess.run([B("bwa"), "mem", "-t", "4", EXAMPLE_FA, f1, f2], env=env, capture_output=True, check=True, timeout=600)
        outs.append(b"\n".join(l for l in r.stdout.split(b"\n") if not l.startswith(b"@PG")))
    assert outs[0].count(b"\n") > 6000 and outs[1] == outs[0]


def test_gpu_deferred_dense_suffix_array(tmp_path, gpu_lib):
    """the denser suffix-array copy made at load time, in the middle of the run, or never: same SAM"""
    fq = str(tmp_path / "r.fq")
    simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 3000, seed=91))
    outs = []
    for after in ("0", "1400", "100000000"):
        env = dict(os.environ, SSG_BWA_DENSIFY_AFTER=after, SSG_BWA_CHUNK_BASES="60000", SSG_BWA_CALL_PAIRS="700")
        r = subprocess.run([B("bwa"), "mem", "-t", "2", "-p", EXAMPLE_FA, fq], capture_output=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(b"\n".join(l for l in r.stdout.split(b"\n") if not l.startswith(b"@PG")))
        assert (b"denser suffix-array copy made after" in r.stderr) == (after == "1400")
    assert outs[0].count(b"\n") > 6000 and outs[1] == outs[0] and outs[2] == outs[0]


def test_gpu_zz_two_calls_in_flight(tmp_path, gpu_lib):
    """SSG_BWA_INFLIGHT=2 and 3 (lanes per device: own streams and arenas, ssg_set_lane; 2 is the default since round 4, when this ran on the
    MI355X for the first time) against one call at a time: same SAM.  A run that does not finish is cut off after two minutes."""
    fq = str(tmp_path / "r.fq")
    simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 4000, seed=93))
    outs = []
    for inflight in ("1", "2", "3"):
        env = dict(os.environ, SSG_BWA_INFLIGHT=inflight, SSG_BWA_CHUNK_BASES="60000", SSG_BWA_CALL_PAIRS="500", SSG_BWA_DENSIFY_AFTER="1500")
        r = subprocess.run([B("bwa"), "mem", "-t", "2", "-p", EXAMPLE_FA, fq], capture_output=True, env=env, timeout=120)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(b"\n".join(l for l in r.stdout.split(b"\n") if not l.startswith(b"@PG")))
    assert outs[0].count(b"\n") > 8000 and outs[1] == outs[0] and outs[2] == outs[0]
