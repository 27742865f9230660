"""K1 + K11 (csrc/k_bam.h, ssg_bam.cpp): FASTQ text in, BAM record bytes out, both ends on the device.

The reference for the bytes is the host formatter the fused path has used since round 3 (csrc/sam_format.cpp ssg_bam_format: htslib-1.3.1
sam.c:835-1028 sam_parse1 typing, sam.c:443-473 bam_write1 layout), itself pinned to the reference's samtools by tests/test_fused.py and
tests/test_sambamba.py.  Here: the device's bytes equal the host formatter's byte for byte, from parsed reads and from raw FASTQ text, over
option letters that change the records (-M, -Y), read names with /1 /2 suffixes and comments, reads without qualities, several upstream batches.
"""
import numpy as np
import pytest

import common
from speedseq_amd import capi

BASES = "ACGTN"


def _inputs(n_pairs, seed, read_len=150, **kw):
    pairs, seqs, seq, off = common.sim_reads(n_pairs, seed, read_len, **kw)
    names = []
    for nm, _, _ in pairs:
        names += [nm, nm]
    rng = np.random.default_rng(seed)
    quals = ["".join(chr(33 + int(q)) for q in rng.integers(2, 41, len(s))) for s in seqs]
    return seqs, seq, off, names, quals


def _fastq(seqs, names, quals, suffix=False, comment=False):
    """interleaved FASTQ text + the offset of every record's '@'"""
    out, rec = [], []
    pos = 0
    for r, (s, nm, q) in enumerate(zip(seqs, names, quals)):
        hdr = "@" + nm + ("/%d" % (1 + (r & 1)) if suffix else "") + (" a comment:%d" % r if comment else "")
        t = "%s\n%s\n+%s\n%s\n" % (hdr, "".join(BASES[c] for c in s), nm if r % 3 == 0 else "", q)
        rec.append(pos); pos += len(t); out.append(t)
    return "".join(out).encode(), np.array(rec, dtype=np.int64)


def _cands_host(res):
    """the fused path's rule for the pairs samblaster may copy to a side stream (host/bwa_main.cpp)"""
    out = []
    for p in range(res.n_pairs):
        nm = [int(np.sum(res.req["kind"][res.req_off[2 * p + i]:res.req_off[2 * p + i + 1]] == 0)) for i in (0, 1)]
        a1, a2 = res.alns[res.req_off[2 * p]], res.alns[res.req_off[2 * p + 1]]
        if nm[0] > 1 or nm[1] > 1 or (a1["rid"] >= 0 and a2["rid"] >= 0 and not (a1["flag"] & 2)):
            out.append((p, nm[0] + nm[1]))
    return out


def _records(b):
    out, p = [], 0
    while p < len(b):
        n = int.from_bytes(b[p:p + 4], "little")
        out.append(b[p:p + 4 + n]); p += 4 + n
    assert p == len(b)
    return out


def check_device_bam(lib, n_pairs, seed, flag=0, read_len=150, rg="grp1", n_batches=1, suffix=False, comment=False, **kw):
    gidx = lib.index_load(common.EXAMPLE_FA)
    seqs, seq, off, names, quals = _inputs(n_pairs, seed, read_len, **kw)
    opt = lib.opt_init()
    opt["flag"] = flag
    pb = (np.arange(n_pairs) * n_batches // n_pairs).astype(np.int32)
    res = capi.mem_process_pairs(lib, gidx, opt, seq, off, pair_batch=pb, n_batches=n_batches, id0=1000)
    want, want_off = capi.bam_format(lib, gidx, opt, res, names, seq, off, quals, rg)
    got = capi.mem_process_pairs_bam(lib, gidx, opt, seq, off, names, quals, pair_batch=pb, n_batches=n_batches, id0=1000, rg_id=rg)
    text, rec_off = _fastq(seqs, names, quals, suffix, comment)
    got2 = capi.mem_process_fastq_bam(lib, gidx, opt, text, rec_off, pair_batch=pb, n_batches=n_batches, id0=1000, rg_id=rg)
    # ... and as two files would arrive: the first reads in one piece, the second reads in another
    t1, r1 = _fastq(seqs[0::2], names[0::2], quals[0::2], suffix, comment)
    t2, r2 = _fastq(seqs[1::2], names[1::2], quals[1::2], suffix, comment)
    rec2 = np.empty(2 * n_pairs, dtype=np.int64); rec2[0::2] = r1; rec2[1::2] = r2 + len(t1)
    got3 = capi.mem_process_fastq_bam(lib, gidx, opt, [t1, t2], rec2, pair_batch=pb, n_batches=n_batches, id0=1000, rg_id=rg)
    for g, what in ((got, "parsed reads"), (got2, "FASTQ text"), (got3, "FASTQ text of two files")):
        if g["bam"] != want:
            a, b = _records(g["bam"]), _records(want)
            assert len(a) == len(b), (what, len(a), len(b))
            for i, (x, y) in enumerate(zip(a, b)):
                assert x == y, (what, i, x, y)
        recs = _records(want)
        assert g["n_rec"] == len(recs)
        hc = _cands_host(res)
        assert [(int(c["pair"]), int(c["n_rec"])) for c in g["cands"]] == hc, what
        first = np.concatenate([[0], np.cumsum([int(np.sum(res.req["kind"][res.req_off[2 * p]:res.req_off[2 * p + 2]] == 0)) for p in range(n_pairs)])])
        boff = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
        for c in g["cands"]:
            assert c["first_rec"] == first[c["pair"]] and c["byte_off"] == boff[c["first_rec"]] and c["n_bytes"] == boff[c["first_rec"] + c["n_rec"]] - boff[c["first_rec"]]
        assert np.array_equal(g["pes"], res.pes) and np.array_equal(g["stats"], res.stats)
    n_sup = sum(1 for r in _records(want) if int.from_bytes(r[18:20], "little") & 0x800)
    n_cand = len(got["cands"])
    res.close()
    lib.index_destroy(gidx)
    return len(_records(want)), n_sup, n_cand


CASES = [dict(n_pairs=300, seed=5), dict(n_pairs=200, seed=6, flag=0x10), dict(n_pairs=200, seed=7, flag=0x200, rg=""),
         dict(n_pairs=240, seed=8, n_batches=3, suffix=True, comment=True, chim_frac=0.2, disc_frac=0.1, n_frac=0.02),
         dict(n_pairs=120, seed=9, read_len=250, ins_mean=800, ins_std=150, chim_frac=0.1)]


@pytest.mark.parametrize("case", CASES)
def test_emu_device_bam_records_equal_the_host_formatter(emu_lib, case):
    n_rec, n_sup, n_cand = check_device_bam(emu_lib, **case)
    assert n_rec >= 2 * case["n_pairs"] and n_cand > 0
    if case.get("chim_frac", 0) >= 0.1:
        assert n_sup > 0


def test_emu_device_bam_without_qualities_and_name_errors(emu_lib):
    lib = emu_lib
    gidx = lib.index_load(common.EXAMPLE_FA)
    seqs, seq, off, names, quals = _inputs(40, 11)
    opt = lib.opt_init()
    res = capi.mem_process_pairs(lib, gidx, opt, seq, off)
    q2 = [q if i % 3 else None for i, q in enumerate(quals)]
    want, _ = capi.bam_format(lib, gidx, opt, res, names, seq, off, q2, "")
    got = capi.mem_process_pairs_bam(lib, gidx, opt, seq, off, names, q2)
    assert got["bam"] == want
    # the two reads of a pair under different names: upstream's words
    text, rec_off = _fastq(seqs, names[:5] + ["other"] + names[6:], quals)
    with pytest.raises(capi.SsgError, match="paired reads have different names: \"%s\", \"other\"" % names[4]):
        capi.mem_process_fastq_bam(lib, gidx, opt, text, rec_off)
    # offsets that are not record starts, a record of several sequence lines: refused, never mis-read
    text, rec_off = _fastq(seqs, names, quals)
    with pytest.raises(capi.SsgError, match="plain four-line record"):
        capi.mem_process_fastq_bam(lib, gidx, opt, text, rec_off + 1)
    bad = text.replace(b"\n+", b"\nACGT\n+", 1)
    with pytest.raises(capi.SsgError, match="plain four-line record"):
        capi.mem_process_fastq_bam(lib, gidx, opt, bad, np.where(np.arange(len(rec_off)) > 0, rec_off + 5, rec_off))
    res.close()
    lib.index_destroy(gidx)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES + [dict(n_pairs=20000, seed=21, n_batches=2, chim_frac=0.05, disc_frac=0.05)])
def test_gpu_device_bam_records_equal_the_host_formatter(gpu_lib, case):
    n_rec, n_sup, n_cand = check_device_bam(gpu_lib, **case)
    assert n_rec >= 2 * case["n_pairs"] and n_cand > 0
