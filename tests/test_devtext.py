"""The device-text path of `bwa mem` (speedseq_amd/host/rawfeed.h; SURVEY.md 2.1 K1 + K11, row f2): with the fused hand-off the input goes to the
device as FASTQ text and comes back as BAM records.  kseq's grammar (/root/reference/src/samtools-1.3.1/htslib-1.3.1/htslib/kseq.h:189-229) accepts
much more than the plain four-line records the device takes -- wrapped sequences, FASTA records, CR LF, blank lines, a quality line that begins
with '@' -- so the host's scanner hands everything from the first upstream batch with such a record to the parser.  Whatever the input, the frames
`bwa mem` writes must be, byte for byte, those of the parser + host formatter path (SSG_BWA_DEVTEXT=0), which tests/test_fused.py ties to the text
path and to the oracle."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

import simreads
from common import EXAMPLE_FA, ROOT

EMU = os.path.join(ROOT, "tests", "emu")
BASES = "ACGTN"


def _records(n_pairs, seed=3, read_len=100):
    contigs = simreads.read_fasta(EXAMPLE_FA)
    pairs = simreads.simulate(contigs, n_pairs, seed=seed, read_len=read_len, chim_frac=0.05, disc_frac=0.05)
    rng = np.random.default_rng(seed)
    out = []
    for nm, r1, r2 in pairs:
        for k, r in enumerate((r1, r2)):
            q = "".join(chr(33 + int(x)) for x in rng.integers(0, 41, len(r)))     # '!' .. 'I': a quality line may begin with '@' or '+'
            out.append(["@%s/%d" % (nm, k + 1), "".join(BASES[c] for c in r), "+", q])
    return out


def _text(recs, eol="\n"):
    return "".join(eol.join(r) + eol for r in recs).encode()


def _run(bwa, args, env_extra, tmp_path, tag):
    env = dict(os.environ, SSG_FUSED="1", SSG_FUSED_SHM="0", SSG_BWA_CHUNK_BASES="3000", SSG_BWA_CALL_PAIRS="40", **env_extra)
    r = subprocess.run([bwa, "mem", "-t", "2", "-R", r"@RG\tID:x\tSM:y"] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    return r.returncode, r.stdout, r.stderr.decode()


def _stream(out):
    """the fused stream without its frame boundaries (fused.h): header text, all BAM record bytes, and per candidate pair (first record counted over the
    whole stream, records, SAM text) -- the parser taking over in the middle of an input starts a device call of its own, and a call is a frame"""
    assert out[:8] == b"SSGFUSE1"
    o, hdr, bam, cands, n_rec, ended = 8, b"", [], [], 0, False
    while o < len(out):
        typ, zero, ln = struct.unpack_from("<IIQ", out, o)
        o += 16
        pay = out[o:o + ln]; o += ln
        if typ == 1:
            hdr = pay
        elif typ == 2:
            nr, nb, nc, nt = struct.unpack_from("<QQQQ", pay, 0)
            cs = [struct.unpack_from("<QQQ", pay, 32 + 24 * k) for k in range(nc)]
            text = pay[32 + 24 * nc:32 + 24 * nc + nt]
            for k, (first, cnt, toff) in enumerate(cs):
                cands.append((n_rec + first, cnt, text[toff:cs[k + 1][2] if k + 1 < nc else nt]))
            bam.append(pay[32 + 24 * nc + nt:]); assert len(bam[-1]) == nb
            n_rec += nr
        elif typ == 4:
            ended = True
        else:
            raise AssertionError("frame type %d" % typ)
    return hdr, b"".join(bam), cands, n_rec, ended


def _both(bwa, args, tmp_path, expect_fallback, expect_fail=False):
    rc0, out0, err0 = _run(bwa, args, {"SSG_BWA_DEVTEXT": "0"}, tmp_path, "parser")
    rc1, out1, err1 = _run(bwa, args, {}, tmp_path, "devtext")
    assert (rc0 != 0) == expect_fail and (rc1 != 0) == expect_fail, (rc0, rc1, err0[-400:], err1[-400:])
    if not expect_fail:
        if not expect_fallback:
            assert out0 == out1, (len(out0), len(out1), err1[-600:])
        a, b = _stream(out0), _stream(out1)
        assert a[4] and b[4] and a[3] > 0
        for k, what in enumerate(("header", "BAM records", "candidate pairs and their SAM text", "record count")):
            assert a[k] == b[k], (what, err1[-600:])
    assert ("the parser takes the rest" in err1) == expect_fallback, err1[-600:]
    assert "the parser takes the rest" not in err0
    return err0, err1


def _variants():
    recs = _records(160)
    v = {}
    v["plain"] = (recs, "\n", False)
    wrapped = [list(r) for r in recs]
    r = wrapped[131]; wrapped[131] = [r[0], r[1][:40], r[1][40:], r[2], r[3][:55], r[3][55:]]          # sequence and qualities over two lines, inside a batch, second read of a pair
    v["wrapped_record"] = (wrapped, "\n", True)
    first = [list(r) for r in recs]
    r = first[0]; first[0] = [r[0], r[1][:10], r[1][10:], r[2], r[3]]
    v["wrapped_first_record"] = (first, "\n", True)
    v["crlf"] = (recs, "\r\n", True)
    blank = [list(r) for r in recs]
    blank[77] = blank[77] + [""]                                                                     # a blank line between two records
    v["blank_line"] = (blank, "\n", False)                                                          # (the one-thread scanner skips it as kseq does: the records stay plain)
    fasta = [list(r) for r in recs]
    fasta[200] = [">" + fasta[200][0][1:], fasta[200][1]]; fasta[201] = [">" + fasta[201][0][1:], fasta[201][1]]   # a pair without qualities
    v["fasta_pair"] = (fasta, "\n", True)
    com = [[r[0] + " 1:N:0:ACGT extra words", r[1], "+" + r[0][1:], r[3]] for r in recs]                # comments and a repeated name on the '+' line: still plain
    v["comments"] = (com, "\n", False)
    v["odd_read_count"] = (recs[:-1], "\n", False)
    return v


VARIANTS = _variants()


@pytest.mark.parametrize("name", sorted(VARIANTS))
@pytest.mark.parametrize("form", ["interleaved", "two_files", "gz_interleaved", "gz_two_files"])
def test_emu_devtext_frames_equal_the_parser_path(tmp_path, name, form):
    recs, eol, fb = VARIANTS[name]
    if name == "odd_read_count" and "two" in form:
        recs = recs + [recs[-1]]      # (two files: the second one a record short instead)
    bwa = os.path.join(EMU, "bwa_emu")
    opn = (lambda p: gzip.open(p, "wb")) if form.startswith("gz") else (lambda p: open(p, "wb"))
    sfx = ".gz" if form.startswith("gz") else ""
    if "two" in form:
        p1, p2 = str(tmp_path / ("r1.fq" + sfx)), str(tmp_path / ("r2.fq" + sfx))
        r2 = recs[1::2][:-1] if name == "odd_read_count" else recs[1::2]
        with opn(p1) as f:
            f.write(_text(recs[0::2], eol))
        with opn(p2) as f:
            f.write(_text(r2, eol))
        args = [EXAMPLE_FA, p1, p2]
    else:
        p1 = str(tmp_path / ("r.fq" + sfx))
        with opn(p1) as f:
            f.write(_text(recs, eol))
        args = ["-p", EXAMPLE_FA, p1]
    err0, err1 = _both(bwa, args, tmp_path, fb)
    if name == "odd_read_count":
        w = "the 2nd file has fewer sequences" if "two" in form else "odd number of reads"
        assert w in err0 and w in err1


def test_emu_devtext_errors_in_upstream_words(tmp_path):
    recs = _records(60)
    bad = [list(r) for r in recs]
    bad[51][0] = "@someone_else/2"
    p = str(tmp_path / "r.fq")
    open(p, "wb").write(_text(bad))
    err0, err1 = _both(os.path.join(EMU, "bwa_emu"), ["-p", EXAMPLE_FA, p], tmp_path, False, expect_fail=True)
    for e in (err0, err1):
        assert '[mem_sam_pe] paired reads have different names: "%s", "someone_else"' % bad[50][0][1:-2] in e, e[-400:]
    trunc = _text(recs)[:-30]                     # the last quality string cut short: kseq's -2
    open(p, "wb").write(trunc)
    err0, err1 = _both(os.path.join(EMU, "bwa_emu"), ["-p", EXAMPLE_FA, p], tmp_path, True, expect_fail=True)
    assert "truncated or malformed FASTQ" in err0 and "truncated or malformed FASTQ" in err1


@pytest.mark.parametrize("name", ["plain", "wrapped_record", "blank_line", "comments"])
@pytest.mark.parametrize("form", ["interleaved", "two_files"])
def test_emu_devtext_scanner_slices_and_windows(tmp_path, name, form):
    """the several-thread scanner over many small slices and windows: the runs of bytes it hands on end at every slice and are copied before a window's buffers are reused"""
    recs, eol, fb = VARIANTS[name]
    bwa = os.path.join(EMU, "bwa_emu")
    if form == "two_files":
        p1, p2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
        open(p1, "wb").write(_text(recs[0::2], eol)); open(p2, "wb").write(_text(recs[1::2], eol))
        args = [EXAMPLE_FA, p1, p2]
    else:
        p1 = str(tmp_path / "r.fq")
        open(p1, "wb").write(_text(recs, eol))
        args = ["-p", EXAMPLE_FA, p1]
    for sl, th in (("700", "3"), ("5000", "2")):
        os.environ["SSG_RANKS_SCAN_SLICE"], os.environ["SSG_RANKS_SCAN_THREADS"] = sl, th
        try:
            _both(bwa, args, tmp_path, fb)
        finally:
            del os.environ["SSG_RANKS_SCAN_SLICE"], os.environ["SSG_RANKS_SCAN_THREADS"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,form", [("plain", "interleaved"), ("plain", "gz_two_files"), ("wrapped_record", "two_files"), ("comments", "gz_interleaved"), ("crlf", "interleaved"), ("odd_read_count", "two_files")])
def test_gpu_devtext_frames_equal_the_parser_path(tmp_path, gpu_lib, name, form):
    """MI355X twin: the same comparison through bin/bwa (FASTQ text unpacked and BAM records written by the gfx950 kernels of csrc/k_bam.h)"""
    recs, eol, fb = VARIANTS[name]
    if name == "odd_read_count":
        recs = recs + [recs[-1]]
    bwa = os.path.join(ROOT, "bin", "bwa")
    opn = (lambda p: gzip.open(p, "wb")) if form.startswith("gz") else (lambda p: open(p, "wb"))
    sfx = ".gz" if form.startswith("gz") else ""
    if "two" in form:
        p1, p2 = str(tmp_path / ("r1.fq" + sfx)), str(tmp_path / ("r2.fq" + sfx))
        with opn(p1) as f:
            f.write(_text(recs[0::2], eol))
        with opn(p2) as f:
            f.write(_text(recs[1::2][:-1] if name == "odd_read_count" else recs[1::2], eol))
        args = [EXAMPLE_FA, p1, p2]
    else:
        p1 = str(tmp_path / ("r.fq" + sfx))
        with opn(p1) as f:
            f.write(_text(recs, eol))
        args = ["-p", EXAMPLE_FA, p1]
    _both(bwa, args, tmp_path, fb)
