#!/usr/bin/env python3
"""The duplicate set (k_sbl.h: open-addressing table in HBM that doubles by re-hash) at whole-genome size: distinct pairs in chunks until the table
has doubled past 2^29 slots, every verdict 0; then the first chunk once more, every verdict 1.  usage: sbl_growth.py [millions of pairs] (MI355X)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from speedseq_amd import capi  # noqa: E402

END_DT = np.dtype([("seq", "i4"), ("pos", "i4"), ("flag", "i4"), ("lclip", "i4"), ("rclip", "i4"), ("ralen", "i4")])


def chunk(first, n):
    e = np.zeros(2 * n, dtype=END_DT)
    k = np.arange(first, first + n, dtype=np.int64)
    e["seq"][0::2] = (k % 24).astype(np.int32); e["seq"][1::2] = e["seq"][0::2]
    e["pos"][0::2] = (k // 24 + 1).astype(np.int32); e["pos"][1::2] = e["pos"][0::2] + 300
    e["flag"][0::2] = 0x63; e["flag"][1::2] = 0x93
    e["ralen"] = 150
    return e


def main():
    total = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 300000000
    step = 8000000
    lib = capi.Lib(None)
    lib.l.ssg_sbl_state_new.restype = C.c_void_p
    st = C.c_void_p(lib.l.ssg_sbl_state_new())
    t0 = time.time(); done = 0
    while done < total:
        n = min(step, total - done)
        e = chunk(done, n); dup = np.zeros(n, dtype=np.uint8)
        rc = lib.l.ssg_sbl_markdup_stream(st, C.c_long(n), e.ctypes.data_as(C.c_void_p), dup.ctypes.data_as(C.c_void_p))
        if rc:
            lib.l.ssg_last_error.restype = C.c_char_p
            print("FAILED at %d pairs: rc %d %s" % (done, rc, lib.l.ssg_last_error())); return 1
        if dup.any():
            print("WRONG at %d pairs: %d of %d distinct pairs called duplicates" % (done, int(dup.sum()), n)); return 1
        done += n
        if done % (40 * 1000000) == 0:
            print("%d M pairs in the set, %.1f s" % (done // 1000000, time.time() - t0), flush=True)
    e = chunk(0, step); dup = np.zeros(step, dtype=np.uint8)
    rc = lib.l.ssg_sbl_markdup_stream(st, C.c_long(step), e.ctypes.data_as(C.c_void_p), dup.ctypes.data_as(C.c_void_p))
    print("first chunk again: rc %d, %d of %d duplicates" % (rc, int(dup.sum()), step))
    lib.l.ssg_sbl_state_free(st)
    return 0 if rc == 0 and dup.all() else 1


if __name__ == "__main__":
    sys.exit(main())
