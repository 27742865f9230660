"""bin/bwa's several-thread reader of plain FASTQ files (speedseq_amd/host/fastq.h, fq_feed_t::parse_plain) against its one-thread
reader on the same bytes: the records and the end state (EOF / truncated) must be the same for ANY input, because every piece runs
the kseq grammar (htslib-1.3.1/htslib/kseq.h:189-229) and a cut that turns out not to be a record start sends the rest of the
file through one thread."""
import os
import random
import subprocess

import pytest

from common import ROOT

FQ_DUMP = os.path.join(ROOT, "tests", "emu", "fq_dump")


def dump(path, threads, piece=None):
    env = dict(os.environ, SSG_FASTQ_THREADS=str(threads), SSG_DEBUG="1")
    if piece:
        env["SSG_FASTQ_PIECE"] = str(piece)
    r = subprocess.run([FQ_DUMP, path], env=env, capture_output=True, check=True, timeout=120)
    return r.stdout, r.stderr.decode()


def four_line(rng, n, nl="\n", qual_chars="@+>ABCDEFGHIJ!~", var_len=True):
    out = []
    for i in range(n):
        l = rng.randint(1, 160) if var_len else 100
        out.append("@r%d%s%s%s%s+%s%s%s" % (i, rng.choice(["", "/1", " c:x", "\tBC:Z:AC"]), nl,
                                             "".join(rng.choices("ACGTNacgt", k=l)), nl,
                                             rng.choice(["", "r%d" % i]) + nl, "".join(rng.choices(qual_chars, k=l)), nl))
    return "".join(out)


def wrapped(rng, n, width):
    """multi-line FASTQ: sequence and quality wrapped at `width`; quality lines begin with '@' and '+' often"""
    out = []
    for i in range(n):
        l = rng.randint(1, 4 * width)
        s = "".join(rng.choices("ACGT", k=l))
        q = "".join(rng.choices("@+I", k=l))
        out.append("@w%d\n" % i)
        out += [s[k:k + width] + "\n" for k in range(0, l, width)]
        out.append("+\n")
        out += [q[k:k + width] + "\n" for k in range(0, l, width)]
    return "".join(out)


CASES = {
    "four_line": lambda rng: four_line(rng, 3000),
    "fixed_len": lambda rng: four_line(rng, 3000, var_len=False),
    "crlf": lambda rng: four_line(rng, 2000, nl="\r\n"),
    "wrapped": lambda rng: wrapped(rng, 1500, 20),
    "wrapped_after_clean": lambda rng: four_line(rng, 1500) + wrapped(rng, 800, 17) + four_line(rng, 500),
    "fasta_mix": lambda rng: four_line(rng, 500) + "".join(">f%d d\n%s\n" % (i, "ACGTTGCA" * rng.randint(1, 9)) for i in range(700)) + four_line(rng, 500),
    "junk_between": lambda rng: "".join(four_line(rng, 1) + rng.choice(["", "\n", "junk line\n", "\n\n"]) for _ in range(2000)),
    "truncated_middle": lambda rng: four_line(rng, 1000) + "@bad\nACGTACGT\n+\nIIII\n" + four_line(rng, 1000),
    "truncated_end": lambda rng: four_line(rng, 2000) + "@bad\nACGTACGT\n+\nIIII",
    "no_final_newline": lambda rng: four_line(rng, 2000)[:-1],
    "empty_reads": lambda rng: "".join(four_line(rng, 1) + rng.choice(["", "@e\n\n+\n\n"]) for _ in range(2000)),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("piece", [257, 4096, 50000])
def test_same_records_as_one_thread(tmp_path, case, piece):
    rng = random.Random(hash((case, piece)) & 0xffff)
    p = tmp_path / "x.fq"
    p.write_text(CASES[case](rng))
    ref, _ = dump(str(p), 1)
    got, _ = dump(str(p), 4, piece)
    assert ref.splitlines()[-1].startswith(b"end\t")
    assert got == ref
    got3, _ = dump(str(p), 3, piece + 13)
    assert got3 == ref


def test_fallback_is_taken_and_reported(tmp_path):
    """a wrapped file whose quality lines look like headers: some cut lands inside a quality string and the reader says so"""
    rng = random.Random(5)
    p = tmp_path / "w.fq"
    out = []
    for i in range(3000):   # every wrapped quality line is a perfect fake record start: '@' line, bases-like line, '+' line, same length
        out.append("@w%d\nACGTACGT\nACGTACGT\nACGTACGT\nACGTACGT\n+\n@CGTACGT\nACGTACGT\n+CGTACGT\nACGTACGT\n" % i)
    p.write_text("".join(out))
    ref, _ = dump(str(p), 1)
    got, err = dump(str(p), 4, 1000)
    assert got == ref
    assert "one thread from there" in err
    assert ref.splitlines()[-1] == b"end\t-1\t3000"


def test_gzip_and_small_files_use_the_serial_reader(tmp_path):
    import gzip
    rng = random.Random(9)
    txt = four_line(rng, 500)
    p = tmp_path / "s.fq"
    p.write_text(txt)
    g = tmp_path / "s.fq.gz"
    with gzip.open(g, "wt") as f:
        f.write(txt)
    a, _ = dump(str(p), 4)            # smaller than two default pieces
    b, _ = dump(str(g), 4, 300)       # gzip magic
    assert a == b


def test_bwa_emu_same_sam_with_pieces(tmp_path, emu_lib):
    """bin/bwa (emulation build) on two plain files: the SAM text does not depend on how the files were cut"""
    import simreads
    from common import EXAMPLE_FA
    bwa = os.path.join(ROOT, "tests", "emu", "bwa_emu")
    contigs = simreads.read_fasta(EXAMPLE_FA)
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    simreads.write_fastq(f1, simreads.simulate(contigs, 400, seed=77), interleaved=False, path2=f2)
    outs = []
    for threads, piece in ((1, None), (4, 3000), (3, 777)):
        env = dict(os.environ, SSG_FASTQ_THREADS=str(threads))
        if piece:
            env["SSG_FASTQ_PIECE"] = str(piece)
        r = subprocess.run([bwa, "mem", "-t", "4", EXAMPLE_FA, f1, f2], env=env, capture_output=True, check=True)
        outs.append(b"\n".join(l for l in r.stdout.split(b"\n") if not l.startswith(b"@PG")))
    assert outs[0].count(b"\n") > 800
    assert outs[1] == outs[0] and outs[2] == outs[0]


@pytest.mark.parametrize("case", ["four_line", "wrapped", "fasta_mix", "truncated_end", "no_final_newline"])
def test_gzip_input_through_the_fast_decoder(tmp_path, case):
    """plain gzip files go through this repository's decoder (host/fast_inflate.h) with the CRC on a thread of its own: same records and end
    state as through zlib (SSG_GZ_FAST=0) and as the uncompressed file; several members; a damaged or cut file is an error, not a short input"""
    import gzip
    rng = random.Random(hash(case) & 0xfff)
    txt = CASES[case](rng).encode()
    p = tmp_path / "x.fq"
    p.write_bytes(txt)
    g = tmp_path / "x.fq.gz"
    g.write_bytes(gzip.compress(txt[:len(txt) // 2], 6) + gzip.compress(txt[len(txt) // 2:], 1))
    ref, _ = dump(str(p), 1)

    def dump_gz(path, fast, threads=1):
        env = dict(os.environ, SSG_GZ_FAST="1" if fast else "0", SSG_GZ_THREADS=str(threads), SSG_GZ_CHUNK="20000")
        return subprocess.run([FQ_DUMP, path], env=env, capture_output=True, check=True, timeout=120).stdout
    assert dump_gz(str(g), True) == ref and dump_gz(str(g), False) == ref
    assert dump_gz(str(g), True, threads=3) == ref                 # several decoding threads for the one stream (fast_inflate_mt.h)
    blob = g.read_bytes()
    cut = tmp_path / "cut.fq.gz"
    dam = bytearray(blob)
    dam[len(dam) // 3] ^= 0x10
    for threads in (1, 3):
        cut.write_bytes(blob[:len(blob) * 2 // 3])
        out = dump_gz(str(cut), True, threads)
        assert out.splitlines()[-1].split(b"\t")[1] == b"-2"      # reported as malformed
        cut.write_bytes(bytes(dam))
        assert dump_gz(str(cut), True, threads).splitlines()[-1].split(b"\t")[1] == b"-2"


def test_bwa_emu_gz_input_several_decoding_threads(tmp_path, emu_lib):
    """bin/bwa on a gz file that spans many chunks of the several-thread decoder: same SAM as with zlib"""
    import gzip
    import simreads
    from common import EXAMPLE_FA
    bwa = os.path.join(ROOT, "tests", "emu", "bwa_emu")
    fq = str(tmp_path / "r.fq")
    simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 1500, seed=79))
    gzp = fq + ".gz"
    with open(fq, "rb") as f, gzip.open(gzp, "wb", 6) as g:
        g.write(f.read())
    outs = []
    for env in ({"SSG_GZ_FAST": "0"}, {"SSG_GZ_THREADS": "1"}, {"SSG_GZ_THREADS": "4", "SSG_GZ_CHUNK": "30000"}):
        r = subprocess.run([bwa, "mem", "-t", "4", "-p", EXAMPLE_FA, gzp], env=dict(os.environ, **env), capture_output=True, check=True, timeout=900)
        outs.append(b"\n".join(l for l in r.stdout.split(b"\n") if not l.startswith(b"@PG")))
    assert outs[0].count(b"\n") > 3000 and outs[1] == outs[0] and outs[2] == outs[0]


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("slab,piece", [(20000, 3000), (400, 150), (90, 40), (1000000, 40000)])
def test_compressed_input_parsed_by_several_threads(tmp_path, case, slab, piece):
    """gz and bgzip input whose decoded stream is cut into slabs and pieces and parsed by several threads (fq_feed_t::parse_stream): same
    records and end state as one thread, for four-line files and for everything else (wrapped, FASTA, junk, truncated: the fall-back to
    one thread happens in the middle of the stream)"""
    import gzip
    rng = random.Random(hash((case, slab)) & 0xffff)
    txt = CASES[case](rng).encode()
    p = tmp_path / "x.fq"
    p.write_bytes(txt)
    g = tmp_path / "x.fq.gz"
    g.write_bytes(gzip.compress(txt, 6))
    ref, _ = dump(str(p), 1)
    env = dict(os.environ, SSG_FASTQ_THREADS="4", SSG_FASTQ_SLAB=str(slab), SSG_FASTQ_PIECE=str(piece), SSG_GZ_THREADS="3", SSG_GZ_CHUNK="30000", SSG_DEBUG="1")
    r = subprocess.run([FQ_DUMP, str(g)], env=env, capture_output=True, check=True, timeout=60)
    assert r.stdout == ref
    env["SSG_GZ_FAST"] = "0"                            # ... and behind zlib's decoder
    assert subprocess.run([FQ_DUMP, str(g)], env=env, capture_output=True, check=True, timeout=60).stdout == ref


def bgzf_blocks(data, block=60000, eof=True):
    """the byte stream bgzip writes (htslib bgzf.c:298-342): independent members of <= 64 KB with their size in a 'BC' extra field"""
    import struct
    import zlib
    out = []
    parts = [data[k:k + block] for k in range(0, len(data), block)] + ([b""] if eof else [])
    for p in parts:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(p) + c.flush()
        size = 12 + 6 + len(body) + 8
        out.append(b"\x1f\x8b\x08\x04" + bytes(4) + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, size - 1) + body + struct.pack("<II", zlib.crc32(p), len(p)))
    return out


def test_bgzf_input_damage_is_an_error_not_a_short_read(tmp_path):
    """a blocked-gzip FASTQ that is cut inside a member, carries a wrong CRC or length, or goes on with a gzip member that is not BGZF must
    fail the read (end state other than a clean end of file): zlib reports all of them, and aligning the first part of the input as if
    it were all of it is the one outcome that must not happen"""
    import gzip
    import struct
    rng = random.Random(77)
    text = four_line(rng, 6000, qual_chars="ABCDEFGHIJ").encode()
    blocks = bgzf_blocks(text)
    assert len(blocks) > 8
    n_all = text.count(b"\n@r")  + 1
    p = tmp_path / "r.fq.gz"

    def end_state(blob, threads):
        p.write_bytes(blob)
        env = dict(os.environ, SSG_BGZF_THREADS=str(threads))
        r = subprocess.run([FQ_DUMP, str(p)], env=env, capture_output=True, check=True, timeout=120)
        last = r.stdout.decode().strip().split("\n")[-1].split("\t")
        return int(last[1]), int(last[2])
    for threads in (1, 3):
        rc, n = end_state(b"".join(blocks), threads)
        assert rc == -1 and n == 6000, (rc, n)                                  # intact: clean end, every record
        assert end_state(b"".join(bgzf_blocks(text, eof=False)), threads) == (-1, 6000)   # no end-of-file block: htslib warns, reads on
        assert end_state(b"".join(blocks) + bytes(100), threads) == (-1, 6000)  # bytes that are no gzip member: ignored, as gzread does
        cut = b"".join(blocks[:5]) + blocks[5][:len(blocks[5]) // 2]
        assert end_state(cut, threads)[0] != -1                                 # cut inside the 6th member
        assert end_state(b"".join(blocks[:5]) + blocks[5][:10], threads)[0] != -1
        bad_crc = bytearray(blocks[3]); bad_crc[-8] ^= 0x40
        assert end_state(b"".join(blocks[:3]) + bytes(bad_crc) + b"".join(blocks[4:]), threads)[0] != -1
        bad_len = bytearray(blocks[3]); bad_len[-4:] = struct.pack("<I", struct.unpack("<I", bytes(bad_len[-4:]))[0] - 1)
        assert end_state(b"".join(blocks[:3]) + bytes(bad_len) + b"".join(blocks[4:]), threads)[0] != -1
        more = four_line(rng, 500, qual_chars="ABCDEFGHIJ").encode()
        assert end_state(b"".join(blocks[:-1]) + gzip.compress(more), threads)[0] != -1   # a plain gzip member behind the BGZF ones
