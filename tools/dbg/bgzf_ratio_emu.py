#!/usr/bin/env python3
"""Compression ratio of the device's BGZF deflate (csrc/k_bgzf.h) on a BAM payload, on the host emulation of the kernel (CPU; for work on the match finder):
usage: bgzf_ratio_emu.py PAYLOAD.bin [N_BLOCKS] [LIB]   -- prints the ratio next to zlib level 1 / 6 and checks every block with zlib's inflate"""
import ctypes as C
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
raw = open(sys.argv[1], "rb").read()
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
lib = C.CDLL(sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "tests", "emu", "libssgpu_emu.so"))
blocks = [raw[k:k + 0xff00] for k in range(0, len(raw), 0xff00)][:nb]
payload = np.frombuffer(b"".join(blocks), dtype=np.uint8)
cut = np.zeros(len(blocks) + 1, dtype=np.uint64); cut[1:] = np.cumsum([len(b) for b in blocks])
cap = int(cut[-1]) + 5 * len(blocks) + 64
out = np.zeros(cap, dtype=np.uint8); off = np.zeros(len(blocks) + 1, dtype=np.uint64)
rc = lib.ssg_bgzf_deflate(payload.ctypes.data_as(C.c_void_p), cut.ctypes.data_as(C.c_void_p), C.c_long(len(blocks)), out.ctypes.data_as(C.c_void_p), C.c_uint64(cap), off.ctypes.data_as(C.c_void_p))
assert rc == 0
tot = 0
for i, b in enumerate(blocks):
    s = out[int(off[i]):int(off[i + 1])].tobytes()
    d = zlib.decompressobj(-15); got = d.decompress(s)
    assert d.eof and got == b, i
    tot += len(s)
z = {}
for lvl in (1, 6):
    z[lvl] = sum(len(zlib.compressobj(lvl, zlib.DEFLATED, -15).compress(b)) + len(zlib.compressobj(lvl, zlib.DEFLATED, -15).flush()) for b in blocks)
    c = 0
    for b in blocks:
        o = zlib.compressobj(lvl, zlib.DEFLATED, -15); c += len(o.compress(b) + o.flush())
    z[lvl] = c
n = sum(len(b) for b in blocks)
print("blocks %d, payload %d: device %.4f   zlib-1 %.4f   zlib-6 %.4f   device / zlib-6 = %.3f" % (len(blocks), n, tot / n, z[1] / n, z[6] / n, tot / z[6]))
