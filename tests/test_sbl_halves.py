"""ssg_sbl_process in two halves (include/ssgpu.h: ssg_sbl_ends -> ssg_sbl_markdup_stream -> ssg_sbl_classify; rank mode's shared duplicate set):
the same line bits and mate lines as the one-call form, chunk after chunk against a persistent set, on random name-grouped blocks with
duplicates, supplementary lines, unmapped ends and unpaired blocks."""
import ctypes as C

import numpy as np
import pytest

from speedseq_amd import capi

LINE_DT = np.dtype([("seq", "i4"), ("pos", "i4"), ("flag", "i4"), ("mapq", "i4"), ("lclip", "i4"), ("rclip", "i4"), ("qalen", "i4"), ("ralen", "i4")])
OPT_DT = np.dtype([("exclude_dups", "i4"), ("add_mate_tags", "i4"), ("max_split_count", "i4"), ("min_non_overlap", "i4"), ("max_unmapped_bases", "i4"), ("min_indel_size", "i4")])


def _blocks(rng, n_blocks):
    lines, off = [], [0]
    for _ in range(n_blocks):
        kind = rng.integers(0, 10)
        def line(flag, mapped=True):
            l = np.zeros((), dtype=LINE_DT)
            l["seq"] = rng.integers(0, 3) if mapped else -1
            l["pos"] = rng.integers(1, 400) if mapped else 0
            l["flag"] = flag | (0 if mapped else 4) | (16 if rng.integers(0, 2) else 0)
            l["mapq"] = rng.integers(0, 61)
            l["lclip"] = rng.integers(0, 3) * 10; l["rclip"] = rng.integers(0, 3) * 10
            l["qalen"] = 100 - int(l["lclip"]) - int(l["rclip"]); l["ralen"] = int(l["qalen"]) + int(rng.integers(-2, 3))
            return l
        if kind == 0:                                   # an unpaired block
            blk = [line(0)]
        else:
            proper = 2 if rng.integers(0, 4) else 0
            blk = [line(0x41 | proper, mapped=kind != 1), line(0x81 | proper, mapped=kind != 2)]
            if kind >= 7:                               # a supplementary line of one of the ends
                blk.insert(1, line(0x841 | proper))
            if kind == 9:
                blk.append(line(0x881 | proper))
        lines += blk
        off.append(len(lines))
    return np.array(lines, dtype=LINE_DT), np.array(off, dtype=np.int64)


def _halves_equal_whole(lib, seed):
    l = lib.l
    l.ssg_sbl_state_new.restype = C.c_void_p
    o = np.zeros((), dtype=OPT_DT)
    l.ssg_sbl_opt_init(capi._ptr(o))
    o["exclude_dups"] = 1; o["add_mate_tags"] = 1; o["max_split_count"] = 2; o["min_non_overlap"] = 20
    st_a, st_b = C.c_void_p(l.ssg_sbl_state_new()), C.c_void_p(l.ssg_sbl_state_new())
    rng = np.random.default_rng(seed)
    n_dup = 0
    for chunk in range(4):                              # the duplicate set persists over the chunks
        lines, off = _blocks(rng, 3000)
        nb, nl = len(off) - 1, len(lines)
        bits_a, mate_a = np.zeros(nl, dtype=np.uint8), np.zeros(nl, dtype=np.int64)
        lib._chk(l.ssg_sbl_process(st_a, capi._ptr(o), C.c_long(nb), capi._ptr(off), capi._ptr(lines), capi._ptr(bits_a), capi._ptr(mate_a)))
        ends = np.zeros(2 * nb, dtype=capi.SBL_END_DT)
        lib._chk(l.ssg_sbl_ends(C.c_long(nb), capi._ptr(off), capi._ptr(lines), capi._ptr(ends)))
        dup = np.zeros(nb, dtype=np.uint8)
        lib._chk(l.ssg_sbl_markdup_stream(st_b, C.c_long(nb), capi._ptr(ends), capi._ptr(dup)))
        bits_b, mate_b = np.zeros(nl, dtype=np.uint8), np.zeros(nl, dtype=np.int64)
        lib._chk(l.ssg_sbl_classify(capi._ptr(o), C.c_long(nb), capi._ptr(off), capi._ptr(lines), capi._ptr(dup), capi._ptr(bits_b), capi._ptr(mate_b)))
        assert np.array_equal(bits_a, bits_b) and np.array_equal(mate_a, mate_b), chunk
        n_dup += int(dup.sum())
    l.ssg_sbl_state_free(st_a); l.ssg_sbl_state_free(st_b)
    return n_dup


def test_emu_sbl_halves_equal_the_whole(emu_lib):
    assert _halves_equal_whole(emu_lib, 5) > 500


@pytest.mark.gpu
def test_gpu_sbl_halves_equal_the_whole(gpu_lib):
    assert _halves_equal_whole(gpu_lib, 6) > 500
