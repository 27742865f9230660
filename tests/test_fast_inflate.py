"""bin/bwa's gzip decoder (speedseq_amd/host/fast_inflate.h) against zlib: same bytes (size + CRC-32 of the output, and the decoder's own
check of each member's CRC / length trailer) for every kind of stream zlib can write, and an error -- not wrong bytes, not a hang -- for
truncated and damaged ones."""
import os
import random
import struct
import subprocess
import zlib

import pytest

from common import ROOT

FI = os.path.join(ROOT, "tests", "emu", "fi_test")
FI_MT = os.path.join(ROOT, "tests", "emu", "fi_mt_test")
# every check runs through the one-thread decoder and through the several-thread one (fast_inflate_mt.h) with chunks small enough that
# the test streams span many of them: block-boundary search, markers for the unknown window, stitching, the one-thread fall-back
MODES = [[FI], [FI_MT, "3", "70000"], [FI_MT, "5", "300000"]]


def gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strategy)
    return c.compress(data) + c.flush()


def run(path, mode=None):
    cmd = [FI, path, "crc"] if not mode or mode[0] == FI else [mode[0], path] + mode[1:]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    return r.returncode, r.stdout, r.stderr


def check(tmp_path, blob, expect, members=None, name="x.gz"):
    import re
    p = tmp_path / name
    p.write_bytes(blob)
    for mode in MODES:
        rc, out, err = run(str(p), mode)
        assert rc == 0, (mode, err)
        m = re.match(r"(\d+) bytes, (\d+) members, crc ([0-9a-f]{8}),", out)
        assert m and int(m.group(1)) == len(expect) and m.group(3) == "%08x" % zlib.crc32(expect), (mode, out)
        if members is not None:
            assert int(m.group(2)) == members, (mode, out)


def fastq(rng, n):
    out = []
    for i in range(n):
        l = rng.randint(50, 150)
        out.append("@r%d/1\n%s\n+\n%s\n" % (i, "".join(rng.choices("ACGTN", k=l)), "".join(rng.choices("IIIIIHHGF#", k=l))))
    return "".join(out).encode()


SAMPLES = {
    "fastq": lambda rng: fastq(rng, 40000),
    "empty": lambda rng: b"",
    "one_byte": lambda rng: b"A",
    "random": lambda rng: bytes(rng.getrandbits(8) for _ in range(300000)),
    "zeros": lambda rng: bytes(3000000),
    "short_periods": lambda rng: b"".join((b"ab" * 700, b"abc" * 500, b"abcdefg" * 300, b"x" * 5000, b"0123456789ABCDE" * 400)) * 20,
    "skewed": lambda rng: bytes(rng.choices(range(256), weights=[2.0 ** -(i % 23) for i in range(256)], k=400000)),   # long and short codes side by side
    "text_far_matches": lambda rng: (fastq(rng, 300) + bytes(rng.getrandbits(8) for _ in range(30000))) * 12,       # matches 30 KB back, across chunks
}


@pytest.mark.parametrize("name", sorted(SAMPLES))
@pytest.mark.parametrize("level", [0, 1, 4, 6, 9])
def test_same_bytes_as_zlib(tmp_path, name, level):
    data = SAMPLES[name](random.Random(hash(name) & 0xffff))
    check(tmp_path, gz(data, level), data, members=1)


@pytest.mark.parametrize("strategy", [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
def test_block_kinds(tmp_path, strategy):
    rng = random.Random(5)
    data = fastq(rng, 20000) + bytes(100000) + bytes(rng.getrandbits(8) for _ in range(50000))
    check(tmp_path, gz(data, 6, strategy), data)
    check(tmp_path, gz(data, 6, strategy, mem=1), data)     # small deflate memory: many short blocks


def test_members_headers_and_trailing_bytes(tmp_path):
    rng = random.Random(9)
    a, b, c = fastq(rng, 5000), b"", fastq(rng, 7000)
    blob = gz(a, 6) + gz(b, 6) + gz(c, 1)
    check(tmp_path, blob, a + b + c, members=3)
    # header with FEXTRA, FNAME, FCOMMENT and FHCRC (RFC 1952): skipped field by field
    body = gz(a, 6)[10:]
    hdr = b"\x1f\x8b\x08" + bytes([4 | 8 | 16 | 2]) + bytes(6) + struct.pack("<H", 5) + b"EXTRA" + b"name.fq\0" + b"a comment\0"
    hdr += struct.pack("<H", zlib.crc32(hdr) & 0xffff)
    check(tmp_path, hdr + body, a, members=1)
    # bytes that are not a member after a complete one are ignored (gzread does the same)
    check(tmp_path, gz(a, 6) + bytes(700), a, members=1)
    # many members, some across the decoder's input refills
    parts = [fastq(rng, rng.randint(1, 3000)) for _ in range(60)]
    check(tmp_path, b"".join(gz(x, rng.choice([1, 6, 9])) for x in parts), b"".join(parts), members=60)


def test_small_members_with_header_fields(tmp_path):
    """header fields on members of a few hundred bytes (the several-thread decoder parses member headers from a buffer of its own: with fewer
    than 8 KB left in the file it used to hand that buffer to the one-thread decoder's refill and crash), alone and behind a large member,
    and a header longer than the first 70 000 bytes read for it"""
    rng = random.Random(21)
    small, big = fastq(rng, 12), fastq(rng, 30000)

    def member(data, flg, extra=b"XY\x03\x00abc", name=b"reads_1.fq\0", comment=b"lane 3\0"):
        hdr = b"\x1f\x8b\x08" + bytes([flg]) + bytes(6)
        if flg & 4:
            hdr += struct.pack("<H", len(extra)) + extra
        if flg & 8:
            hdr += name
        if flg & 16:
            hdr += comment
        if flg & 2:
            hdr += struct.pack("<H", zlib.crc32(hdr) & 0xffff)
        return hdr + gz(data, 6)[10:]
    for flg in (4, 8, 16, 2, 4 | 8, 4 | 8 | 16 | 2):
        check(tmp_path, member(small, flg), small, members=1)
        check(tmp_path, gz(big, 6) + member(small, flg), big + small, members=2)
        check(tmp_path, member(small, flg) + member(small, flg) + gz(small, 1), small * 3, members=3)
    long_name = bytes(rng.choice(b"abcdefghij") for _ in range(200000)) + b"\0"
    check(tmp_path, member(big, 8, name=long_name), big, members=1)
    check(tmp_path, member(small, 4 | 8, extra=bytes(65535), name=long_name), small, members=1)
    p = tmp_path / "cut.gz"                          # the file ends inside a header field: an error, not a crash
    blob = member(small, 4 | 8)
    for cut in (11, 13, 16, 20, 24):
        p.write_bytes(blob[:cut])
        for mode in MODES:
            rc, out, err = run(str(p), mode)
            assert rc == 1 and "error" in err, (mode, cut, out, err)


def test_large_stream_many_chunks(tmp_path):
    rng = random.Random(11)
    data = fastq(rng, 200000)                        # ~50 MB of text: a dozen output chunks, input refills
    check(tmp_path, gz(data, 6), data, members=1)


def test_truncated_and_damaged_streams_are_errors(tmp_path):
    rng = random.Random(13)
    data = fastq(rng, 30000)
    blob = gz(data, 6)
    p = tmp_path / "t.gz"
    for cut in [1, 5, 10, 11, 50, len(blob) // 3, len(blob) // 2, len(blob) - 9, len(blob) - 8, len(blob) - 1]:
        p.write_bytes(blob[:cut])
        for mode in MODES:
            rc, out, err = run(str(p), mode)
            assert rc == 1 and "error" in err, (mode, cut, out, err)
    stored = gz(data, 0)
    for cut in [12, 14, 40000, len(stored) - 3]:
        p.write_bytes(stored[:cut])
        for mode in MODES:
            assert run(str(p), mode)[0] == 1, (mode, cut)
    bad = 0
    for k in range(40):                               # a flipped byte is a structural error or a CRC / length mismatch -- never silence
        pos = rng.randrange(12, len(blob) - 8)
        dam = bytearray(blob)
        dam[pos] ^= 1 << rng.randrange(8)
        p.write_bytes(bytes(dam))
        for mode in MODES:
            rc, out, err = run(str(p), mode)
            assert rc == 1, (mode, pos, out)
        bad += 1
    assert bad == 40
    p.write_bytes(b"not gzip at all")
    assert run(str(p))[0] == 1
    tr = bytearray(blob)
    tr[-1] ^= 0x40                                    # length field of the trailer
    p.write_bytes(bytes(tr))
    assert run(str(p))[0] == 1
    tr = bytearray(blob)
    tr[-6] ^= 0x40                                    # CRC field of the trailer
    p.write_bytes(bytes(tr))
    assert run(str(p))[0] == 1
