#!/usr/bin/env python3
"""BGZF deflate on the device (ssg_bgzf_deflate, k_bgzf.h): rate from and to page-locked host memory, size against zlib level 6 and 1,
and every 50th block inflated by zlib.  usage: bgzf_bench.py [MB of BAM-shaped payload]"""
import ctypes as C
import os
import random
import struct
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from speedseq_amd import capi  # noqa: E402
import test_bgzf_device as T  # noqa: E402


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    lib = capi.Lib(None)
    lib.l.ssg_host_alloc.restype = C.c_void_p
    rng = random.Random(11)
    unit = T.bam_like(rng, 30000)                      # ~10 MB of records, repeated with the names re-drawn by a cheap byte shuffle of the tail
    reps = max(1, mb * 1000000 // len(unit))
    data = np.tile(np.frombuffer(unit, dtype=np.uint8), reps)
    n = data.size
    nb = (n + 0xff00 - 1) // 0xff00
    cut = np.minimum(np.arange(nb + 1, dtype=np.uint64) * 0xff00, n).astype(np.uint64)
    P = lib.l.ssg_host_alloc(C.c_size_t(n + 64)); O = lib.l.ssg_host_alloc(C.c_size_t(n + 5 * nb + 64))
    C.memmove(P, data.ctypes.data, n)
    off = np.zeros(nb + 1, dtype=np.uint64)
    for it in range(3):
        t0 = time.perf_counter()
        rc = lib.l.ssg_bgzf_deflate(C.c_void_p(P), cut.ctypes.data_as(C.c_void_p), C.c_long(nb), C.c_void_p(O), C.c_uint64(n + 5 * nb + 64), off.ctypes.data_as(C.c_void_p))
        dt = time.perf_counter() - t0
        assert rc == 0, lib.l.ssg_last_error()
        print("run %d: %d blocks, %.1f MB -> %.1f MB (ratio %.3f) in %.3f s = %.2f GB/s of payload" % (it, nb, n / 1e6, int(off[-1]) / 1e6, int(off[-1]) / n, dt, n / dt / 1e9))
    out = np.ctypeslib.as_array(C.cast(O, C.POINTER(C.c_uint8)), shape=(int(off[-1]),))
    bad = 0; z6 = z1 = zin = 0
    for b in range(0, nb, 50):
        pay = data[int(cut[b]):int(cut[b + 1])].tobytes()
        d = zlib.decompressobj(-15)
        if d.decompress(out[int(off[b]):int(off[b + 1])].tobytes()) != pay or not d.eof:
            bad += 1
        for lvl in (6, 1):
            c = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8); k = len(c.compress(pay) + c.flush())
            if lvl == 6: z6 += k
            else: z1 += k
        zin += len(pay)
    print("sampled blocks that do not inflate to their payload: %d; zlib on the same sample: level 6 ratio %.3f, level 1 ratio %.3f" % (bad, z6 / zin, z1 / zin))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
