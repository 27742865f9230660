"""BGZF deflate on the device (k_bgzf.h, ssg_bgzf_deflate): every block's stream must be a valid RFC 1951 stream that
zlib inflates to exactly the payload -- for BAM-shaped data, text, runs, noise, empty and one-byte blocks, and blocks of the largest size
bgzf_write cuts (0xff00).  CPU-side on the host emulation of the kernel; `-m gpu` on the MI355X."""
import ctypes as C
import os
import random
import struct
import zlib

import numpy as np
import pytest


def bam_like(rng, n):
    recs = []
    for _ in range(n):
        name = ("read%d" % rng.randrange(10 ** 7)).encode() + b"\0"
        core = struct.pack("<iiIIiiii", rng.randrange(25), rng.randrange(10 ** 8), 0x12345678, (99 << 16) | 1, 150, rng.randrange(25), rng.randrange(10 ** 8), rng.randrange(-500, 500))
        body = core + name + struct.pack("<I", 150 << 4) + bytes(rng.getrandbits(8) for _ in range(75)) + bytes(rng.choice([40, 40, 40, 37, 12]) for _ in range(150)) + b"NMC\x00MDZ150\x00ASC\x96XSC\x00RGZgrp1\x00MCZ150M\x00MQC\x3c"
        recs.append(struct.pack("<I", len(body)) + body)
    return b"".join(recs)


def payloads(seed, big):
    rng = random.Random(seed)
    out = [b"", b"A", b"AB", b"ABC", b"ABCD", b"abcabcabcabc", bytes(1000), bytes([7]) * 0xff00, bam_like(rng, 40)]
    out.append(bytes(rng.getrandbits(8) for _ in range(5000)))                      # noise: stored
    out.append(bytes(rng.getrandbits(8) for _ in range(0xff00)) if big else bytes(rng.getrandbits(8) for _ in range(3000)))
    out.append(("".join(rng.choice(["@r%d/1\n" % rng.randrange(999), "ACGTTGCA" * rng.randint(1, 12) + "\n", "+\n", "IIIIHHHGG#" * rng.randint(1, 9) + "\n"]) for _ in range(900))).encode()[:0xff00])
    out.append(bytes(rng.choices(range(256), weights=[2.0 ** -(i % 29) for i in range(256)], k=60000)))   # skewed: deep Huffman trees
    out.append(bytes([rng.randrange(4)]) * 300 + b"xyz" * 5000 + bytes(range(256)) * 40)
    d = bam_like(rng, 1300 if big else 60)
    out += [d[k:k + 0xff00] for k in range(0, len(d), 0xff00)]
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    deep = b"".join(bytes([i]) * f for i, f in enumerate(fib))[:0xff00]           # Fibonacci frequencies: a tree deeper than 15 bits before limiting
    out.append(bytes(rng.sample(list(deep), len(deep))) if big else deep[:4000])
    return out


def check(lib, seed, big):
    blocks = payloads(seed, big)
    payload = np.frombuffer(b"".join(blocks), dtype=np.uint8)
    cut = np.zeros(len(blocks) + 1, dtype=np.uint64)
    cut[1:] = np.cumsum([len(b) for b in blocks])
    cap = int(cut[-1]) + 5 * len(blocks) + 64
    out = np.zeros(cap, dtype=np.uint8)
    off = np.zeros(len(blocks) + 1, dtype=np.uint64)
    rc = lib.l.ssg_bgzf_deflate(payload.ctypes.data_as(C.c_void_p) if payload.size else None, cut.ctypes.data_as(C.c_void_p), C.c_long(len(blocks)),
                                out.ctypes.data_as(C.c_void_p), C.c_uint64(cap), off.ctypes.data_as(C.c_void_p))
    assert rc == 0, lib.l.ssg_last_error()
    tot_in = tot_out = 0
    for i, b in enumerate(blocks):
        stream = out[int(off[i]):int(off[i + 1])].tobytes()
        assert len(stream) <= len(b) + 5, (i, len(stream), len(b))
        d = zlib.decompressobj(-15)
        got = d.decompress(stream)
        assert d.eof and d.unused_data == b"", (i, len(b))          # one final block, nothing behind it
        assert got == b, (i, len(b), len(got))
        if len(b) > 2000:
            tot_in += len(b); tot_out += len(stream)
    return tot_out / max(1, tot_in)


def test_emu_bgzf_deflate_blocks_inflate_to_their_payload(emu_lib):
    ratio = check(emu_lib, 5, big=False)
    assert ratio < 0.9


@pytest.mark.gpu
def test_gpu_bgzf_deflate_blocks_inflate_to_their_payload(gpu_lib):
    for seed in (5, 6, 7):
        assert check(gpu_lib, seed, big=True) < 0.9
