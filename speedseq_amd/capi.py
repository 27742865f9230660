"""ctypes binding of libssgpu.so's C ABI (include/ssgpu.h).

The product library is the HIP build in this directory.  There is no CPU fallback: if the shared
object is missing or no GPU is visible the calls raise.  Tests may load the host-emulation build of
the same sources (tests/emu/libssgpu_emu.so) by passing its path explicitly.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libssgpu.so")

# numpy mirrors of speedseq_amd/csrc/ssg_types.h
OPT_DT = np.dtype([
    ("a", "i4"), ("b", "i4"), ("o_del", "i4"), ("e_del", "i4"), ("o_ins", "i4"), ("e_ins", "i4"),
    ("pen_unpaired", "i4"), ("pen_clip5", "i4"), ("pen_clip3", "i4"), ("w", "i4"), ("zdrop", "i4"),
    ("T", "i4"), ("min_seed_len", "i4"), ("min_chain_weight", "i4"), ("max_chain_extend", "i4"),
    ("split_width", "i4"), ("max_occ", "i4"), ("max_chain_gap", "i4"), ("max_ins", "i4"), ("max_matesw", "i4"),
    ("max_XA_hits", "i4"), ("max_XA_hits_alt", "i4"), ("mapQ_coef_fac", "i4"), ("chunk_size", "i4"), ("n_threads", "i4"),
    ("_align", "i4"),
    ("max_mem_intv", "u8"),
    ("split_factor", "f4"), ("mask_level", "f4"), ("drop_ratio", "f4"), ("XA_drop_ratio", "f4"),
    ("mask_level_redun", "f4"), ("mapQ_coef_len", "f4"),
    ("mat", "i1", (25,)), ("_pad0", "i1"), ("flag", "u2"), ("_pad", "i1", (4,)),
], align=False)
INTV_DT = np.dtype([("x0", "u8"), ("x1", "u8"), ("x2", "u8"), ("info", "u8")])
SEED_DT = np.dtype([("rbeg", "i8"), ("qbeg", "i4"), ("len", "i4"), ("score", "i4"), ("next", "i4")])
EXT_JOB_DT = np.dtype([("qoff", "i4"), ("qlen", "i4"), ("toff", "i4"), ("tlen", "i4"), ("w", "i4"), ("end_bonus", "i4"), ("zdrop", "i4"), ("h0", "i4")])
EXT_RES_DT = np.dtype([("score", "i4"), ("qle", "i4"), ("tle", "i4"), ("gtle", "i4"), ("gscore", "i4"), ("max_off", "i4")])
SW_JOB_DT = np.dtype([("qoff", "i4"), ("qlen", "i4"), ("toff", "i4"), ("tlen", "i4"), ("xtra", "i4"), ("_pad", "i4")])
KSWR_DT = np.dtype([("score", "i4"), ("te", "i4"), ("qe", "i4"), ("score2", "i4"), ("te2", "i4"), ("tb", "i4"), ("qb", "i4")])
GLB_JOB_DT = np.dtype([("qoff", "i4"), ("qlen", "i4"), ("toff", "i4"), ("tlen", "i4"), ("w", "i4"), ("_pad", "i4")])
ALNREG_DT = np.dtype([
    ("rb", "i8"), ("re", "i8"), ("qb", "i4"), ("qe", "i4"), ("rid", "i4"), ("score", "i4"), ("truesc", "i4"), ("sub", "i4"),
    ("alt_sc", "i4"), ("csub", "i4"), ("sub_n", "i4"), ("w", "i4"), ("seedcov", "i4"), ("secondary", "i4"),
    ("secondary_all", "i4"), ("seedlen0", "i4"), ("n_comp", "i4"), ("frac_rep", "f4"), ("hash", "u8"),
])
PESTAT_DT = np.dtype([("low", "i4"), ("high", "i4"), ("failed", "i4"), ("_pad", "i4"), ("avg", "f8"), ("std", "f8")])


class SsgError(RuntimeError):
    pass


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Lib:
    def __init__(self, path=None):
        path = path or os.environ.get("SSGPU_LIB") or DEFAULT_LIB   # SSGPU_LIB: e.g. the instrumented build of `make tune`
        if not os.path.exists(path):
            raise SsgError("%s not found: build it with `make lib` (hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        self.path = path
        self.l = C.CDLL(path)
        self.l.ssg_version.restype = C.c_char_p
        self.l.ssg_backend.restype = C.c_char_p
        self.l.ssg_last_error.restype = C.c_char_p
        self.l.ssg_index_l_pac.restype = C.c_int64
        assert OPT_DT.itemsize == self.sizeof_opt(), (OPT_DT.itemsize, self.sizeof_opt())

    def sizeof_opt(self):
        # ssg_mem_opt_t: 25 int32 (+4 pad) + u64 + 6 floats + 32 bytes
        return 25 * 4 + 4 + 8 + 6 * 4 + 32

    def _chk(self, rc):
        if rc != 0:
            raise SsgError("libssgpu error %d: %s" % (rc, self.l.ssg_last_error().decode()))

    def backend(self):
        return self.l.ssg_backend().decode()

    def device_count(self):
        return self.l.ssg_device_count()

    def opt_init(self):
        o = np.zeros(1, dtype=OPT_DT)
        self.l.ssg_mem_opt_init(_ptr(o))
        return o

    # ---- index ----
    def index_load(self, prefix):
        h = C.c_void_p()
        self._chk(self.l.ssg_index_load(prefix.encode(), C.byref(h)))
        return h

    def index_build_fasta(self, fasta):
        """upstream `bwa index` on the GPU (ssg_index_build_fasta); returns the index handle."""
        h = C.c_void_p()
        self._chk(self.l.ssg_index_build_fasta(fasta.encode(), C.byref(h)))
        return h

    def index_build_dev(self, d_fwd, l_pac, ctg_off, ctg_len, names=None):
        """index of forward-strand nt4 codes resident in HBM (d_fwd = device address)."""
        h = C.c_void_p()
        co = np.asarray(ctg_off, dtype=np.int64)
        cl = np.asarray(ctg_len, dtype=np.int32)
        self._chk(self.l.ssg_index_build_dev(C.c_void_p(d_fwd), C.c_int64(l_pac), C.c_int(len(co)), _ptr(co), _ptr(cl), C.byref(h)))
        if names is not None:
            arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
            self._chk(self.l.ssg_index_set_names(h, C.c_int(len(names)), arr))
        return h

    def index_save(self, h, prefix):
        self._chk(self.l.ssg_index_save(h, prefix.encode()))

    def index_destroy(self, h):
        self.l.ssg_index_destroy(h)

    # ---- stage-level ----
    def extend_batch(self, opt, jobs, qbuf, tbuf):
        jobs = np.ascontiguousarray(jobs, dtype=EXT_JOB_DT)
        res = np.zeros(len(jobs), dtype=EXT_RES_DT)
        cells = C.c_uint64(0)
        self._chk(self.l.ssg_extend_batch(_ptr(opt), C.c_int(len(jobs)), _ptr(jobs), _ptr(qbuf), C.c_size_t(qbuf.size),
                                          _ptr(tbuf), C.c_size_t(tbuf.size), _ptr(res), C.byref(cells)))
        return res, cells.value

    def align2_batch(self, opt, jobs, qbuf, tbuf):
        jobs = np.ascontiguousarray(jobs, dtype=SW_JOB_DT)
        res = np.zeros(len(jobs), dtype=KSWR_DT)
        self._chk(self.l.ssg_align2_batch(_ptr(opt), C.c_int(len(jobs)), _ptr(jobs), _ptr(qbuf), C.c_size_t(qbuf.size),
                                          _ptr(tbuf), C.c_size_t(tbuf.size), _ptr(res)))
        return res

    def global_batch(self, opt, jobs, qbuf, tbuf, cap=64):
        jobs = np.ascontiguousarray(jobs, dtype=GLB_JOB_DT)
        n = len(jobs)
        score = np.zeros(n, dtype=np.int32)
        ncig = np.zeros(n, dtype=np.int32)
        cig = np.zeros((n, cap), dtype=np.uint32)
        self._chk(self.l.ssg_global_batch(_ptr(opt), C.c_int(n), _ptr(jobs), _ptr(qbuf), C.c_size_t(qbuf.size),
                                          _ptr(tbuf), C.c_size_t(tbuf.size), _ptr(score), _ptr(ncig), _ptr(cig), C.c_int(cap)))
        return score, ncig, cig

    def smem_batch(self, idx, opt, seq, off, cap=64):
        n = len(off) - 1
        intv = np.zeros((n, cap), dtype=INTV_DT)
        cnt = np.zeros(n, dtype=np.int32)
        self._chk(self.l.ssg_smem_batch(idx, _ptr(opt), C.c_int(n), _ptr(seq), _ptr(off), C.c_int(cap), _ptr(intv), _ptr(cnt)))
        return intv, cnt

    def seeds_batch(self, idx, opt, seq, off):
        """seeds of every read before chaining: (seed_off, seeds[rbeg qbeg len score next], rids)"""
        n = len(off) - 1
        seed_off = np.zeros(n + 1, dtype=np.int64)
        sp, rp = C.c_void_p(), C.c_void_p()
        self._chk(self.l.ssg_seeds_batch(idx, _ptr(opt), C.c_int(n), _ptr(seq), _ptr(off), _ptr(seed_off), C.byref(sp), C.byref(rp)))
        tot = int(seed_off[n])
        seeds = np.frombuffer((C.c_char * (tot * SEED_DT.itemsize)).from_address(sp.value), dtype=SEED_DT, count=tot).copy() if tot else np.zeros(0, SEED_DT)
        rids = np.frombuffer((C.c_char * (tot * 4)).from_address(rp.value), dtype=np.int32, count=tot).copy() if tot else np.zeros(0, np.int32)
        self.l.ssg_free(sp); self.l.ssg_free(rp)
        return seed_off, seeds, rids

    def extend_lane_batch(self, idx, opt, jobs, tpos, direction, qbuf, qcap):
        res = np.zeros(len(jobs), dtype=EXT_RES_DT)
        cells = C.c_uint64(0)
        tpos = np.ascontiguousarray(tpos, dtype=np.int64)
        self._chk(self.l.ssg_extend_lane_batch(idx, _ptr(opt), C.c_int(len(jobs)), _ptr(jobs), _ptr(tpos), C.c_int(direction), _ptr(qbuf), C.c_size_t(qbuf.size), C.c_int(qcap), _ptr(res), C.byref(cells)))
        return res, cells.value

    def align2_lane_batch(self, idx, opt, jobs, tpos, qbuf, lanes):
        jobs = np.ascontiguousarray(jobs, dtype=SW_JOB_DT)
        res = np.zeros(len(jobs), dtype=KSWR_DT)
        from_lane = np.zeros(len(jobs), dtype=np.int32)
        tpos = np.ascontiguousarray(tpos, dtype=np.int64)
        self._chk(self.l.ssg_align2_lane_batch(idx, _ptr(opt), C.c_int(len(jobs)), _ptr(jobs), _ptr(tpos), _ptr(qbuf), C.c_size_t(qbuf.size), C.c_int(lanes), _ptr(res), _ptr(from_lane)))
        return res, from_lane

    def dbg_chain_sort(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        o0, o1 = np.zeros(len(keys), dtype=np.int64), np.zeros(len(keys), dtype=np.int64)
        self._chk(self.l.ssg_dbg_chain_sort(_ptr(keys), C.c_int(len(keys)), _ptr(o0), _ptr(o1)))
        return o0, o1

    def align1_batch(self, idx, opt, seq, off):
        n = len(off) - 1
        reg_off = np.zeros(n + 1, dtype=np.int64)
        regs_p = C.c_void_p()
        stats = np.zeros(8, dtype=np.uint64)
        self._chk(self.l.ssg_align1_batch(idx, _ptr(opt), C.c_int(n), _ptr(seq), _ptr(off), _ptr(reg_off), C.byref(regs_p), _ptr(stats)))
        tot = int(reg_off[n])
        buf = (C.c_char * (tot * ALNREG_DT.itemsize)).from_address(regs_p.value) if tot else b""
        regs = np.frombuffer(buf, dtype=ALNREG_DT, count=tot).copy()
        self.l.ssg_free(regs_p)
        return reg_off, regs, stats


ALNREQ_DT = np.dtype([("read", "i4"), ("reg", "i4"), ("kind", "i4"), ("owner", "i4"), ("flag", "i4"), ("mapq", "i4"), ("_p0", "i4"), ("_p1", "i4")])
ALN_DT = np.dtype([("pos", "i8"), ("rid", "i4"), ("flag", "i4"), ("mapq", "i4"), ("NM", "i4"), ("score", "i4"), ("sub", "i4"), ("n_cigar", "i4"),
                   ("is_rev", "i4"), ("l_md", "i4"), ("reg_idx", "i4"), ("xa_cnt", "i4"), ("_pad", "i4"), ("cigar", "u4", (64,)), ("md", "S320")])


class PeResult:
    """Owner of an ssg_pe_result_t (records of one ssg_mem_process_pairs call)."""

    def __init__(self, lib, handle, n_pairs, n_batches):
        self.lib, self.h, self.n_pairs, self.n_batches = lib, handle, n_pairs, n_batches
        l = lib.l
        n = l.ssg_pe_n_req(handle)
        self.req_off = np.ctypeslib.as_array(C.cast(l.ssg_pe_req_off(handle), C.POINTER(C.c_int64)), shape=(2 * n_pairs + 1,)).copy()
        self.req = np.frombuffer((C.c_char * (n * ALNREQ_DT.itemsize)).from_address(l.ssg_pe_req(handle)), dtype=ALNREQ_DT, count=n).copy() if n else np.zeros(0, ALNREQ_DT)
        self.alns = np.frombuffer((C.c_char * (n * ALN_DT.itemsize)).from_address(l.ssg_pe_alns(handle)), dtype=ALN_DT, count=n).copy() if n else np.zeros(0, ALN_DT)
        self.pes = np.frombuffer((C.c_char * (n_batches * 4 * PESTAT_DT.itemsize)).from_address(l.ssg_pe_pes(handle)), dtype=PESTAT_DT, count=n_batches * 4).copy() if n_batches else np.zeros(0, PESTAT_DT)
        self.stats = np.ctypeslib.as_array(C.cast(l.ssg_pe_stats(handle), C.POINTER(C.c_uint64)), shape=(8,)).copy()

    def close(self):
        if self.h:
            self.lib.l.ssg_pe_result_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def _bind_pe(lib):
    l = lib.l
    l.ssg_pe_n_req.restype = C.c_int64
    for f in ("ssg_pe_req_off", "ssg_pe_req", "ssg_pe_alns", "ssg_pe_pes", "ssg_pe_stats"):
        getattr(l, f).restype = C.c_void_p
        getattr(l, f).argtypes = [C.c_void_p]
    l.ssg_pe_n_req.argtypes = [C.c_void_p]
    l.ssg_pe_result_free.argtypes = [C.c_void_p]


def mem_process_pairs(lib, idx, opt, seq, off, pair_batch=None, n_batches=1, id0=0, pes0=None):
    _bind_pe(lib)
    n_pairs = (len(off) - 1) // 2
    if pair_batch is None:
        pair_batch = np.zeros(n_pairs, dtype=np.int32)
    pair_batch = np.ascontiguousarray(pair_batch, dtype=np.int32)
    h = C.c_void_p()
    p0 = _ptr(pes0) if pes0 is not None else None
    lib._chk(lib.l.ssg_mem_process_pairs(idx, _ptr(opt), C.c_int(n_pairs), _ptr(seq), _ptr(off), _ptr(pair_batch), C.c_int(n_batches),
                                         C.c_int64(id0), p0, C.byref(h)))
    return PeResult(lib, h, n_pairs, n_batches)


SBL_END_DT = np.dtype([("seq", "i4"), ("pos", "i4"), ("flag", "i4"), ("lclip", "i4"), ("rclip", "i4"), ("ralen", "i4")])


def sbl_markdup(lib, ends):
    """ends: SBL_END_DT array of 2*n_pairs primary records -> uint8 dup flag per pair."""
    ends = np.ascontiguousarray(ends, dtype=SBL_END_DT)
    n = len(ends) // 2
    dup = np.zeros(n, dtype=np.uint8)
    lib._chk(lib.l.ssg_sbl_markdup(C.c_long(n), _ptr(ends), _ptr(dup)))
    return dup


def index_from_device(lib, d_bwt, primary, L2, d_sa, d_pac, l_pac, ctg_off, ctg_len, sa_intv=32):
    """d_* are integer device addresses (e.g. torch.Tensor.data_ptr()); the caller keeps the tensors alive."""
    h = C.c_void_p()
    L2a = np.asarray(L2, dtype=np.uint64)
    co = np.asarray(ctg_off, dtype=np.int64)
    cl = np.asarray(ctg_len, dtype=np.int32)
    lib._chk(lib.l.ssg_index_from_device(C.c_void_p(d_bwt), C.c_uint64(primary), _ptr(L2a), C.c_void_p(d_sa), C.c_int(sa_intv),
                                         C.c_void_p(d_pac), C.c_int64(l_pac), C.c_int(len(co)), _ptr(co), _ptr(cl), C.byref(h)))
    return h


def hotpath_dev(lib, idx, opt, n_pairs, max_len, d_seq, d_off, d_pair_batch, n_batches=1, id0=0, want_dup=False):
    summary = np.zeros(8, dtype=np.uint64)
    dup = np.zeros(n_pairs, dtype=np.uint8) if want_dup else None
    lib._chk(lib.l.ssg_hotpath_dev(idx, _ptr(opt), C.c_int(n_pairs), C.c_int(max_len), C.c_void_p(d_seq), C.c_void_p(d_off), C.c_void_p(d_pair_batch),
                                   C.c_int(n_batches), C.c_int64(id0), _ptr(summary), _ptr(dup) if want_dup else None))
    return summary, dup


def hotpath_dev_ex(lib, idx, opt, n_pairs, max_len, d_seq, d_off, d_pair_batch, n_batches=1, id0=0, local_dedup=True, d_sig=None, keep=False):
    """ssg_hotpath_dev_ex: summary[16] (see include/ssgpu.h) and, with keep=True, the handle of the records left in HBM."""
    summary = np.zeros(16, dtype=np.uint64)
    h = C.c_void_p()
    lib._chk(lib.l.ssg_hotpath_dev_ex(idx, _ptr(opt), C.c_int(n_pairs), C.c_int(max_len), C.c_void_p(d_seq), C.c_void_p(d_off), C.c_void_p(d_pair_batch),
                                      C.c_int(n_batches), C.c_int64(id0), None, C.c_int(1 if local_dedup else 0), _ptr(summary), None,
                                      C.c_void_p(d_sig) if d_sig else None, C.byref(h) if keep else None))
    return summary, (h if keep else None)


def dev_records_classify(lib, h, d_dup):
    """samblaster's classification of kept records with externally decided duplicate flags (device pointer, one byte per pair)."""
    counts = np.zeros(4, dtype=np.uint64)
    lib._chk(lib.l.ssg_dev_records_classify(h, None, C.c_void_p(d_dup), _ptr(counts)))
    return counts


def markdup_sig_dev(lib, n, d_sig, d_ordinal, d_dup):
    """owner-side first-seen-wins over received signatures (device addresses): ssg_markdup_sig_dev"""
    lib._chk(lib.l.ssg_markdup_sig_dev(C.c_long(n), C.c_void_p(d_sig), C.c_void_p(d_ordinal), C.c_void_p(d_dup)))


def dev_records_n_lines(lib, h):
    lib.l.ssg_dev_records_n_lines.restype = C.c_int64
    return int(lib.l.ssg_dev_records_n_lines(h))


def dev_record_bytes(lib):
    lib.l.ssg_dev_record_bytes.restype = C.c_size_t
    return int(lib.l.ssg_dev_record_bytes())


def dev_records_export(lib, h, d_keys, d_recs=None, d_bits=None):
    """sort keys / fixed-size records / side-stream bits of the kept records into device buffers (addresses)"""
    lib._chk(lib.l.ssg_dev_records_export(h, C.c_void_p(d_keys), C.c_void_p(d_recs) if d_recs else None, C.c_void_p(d_bits) if d_bits else None))


def dev_records_download(lib, h, n_pairs):
    """records kept in HBM by hotpath_dev_ex(keep=True) -> (PeResult, per-line SSG_SBL_* bits, per-line mate line index)"""
    _bind_pe(lib)
    nl = dev_records_n_lines(lib, h)
    bits = np.zeros(nl, dtype=np.uint8)
    mate = np.zeros(nl, dtype=np.int64)
    r = C.c_void_p()
    lib._chk(lib.l.ssg_dev_records_download(h, C.byref(r), _ptr(bits), _ptr(mate)))
    return PeResult(lib, r, n_pairs, 0), bits, mate


def dev_records_free(lib, h):
    lib.l.ssg_dev_records_free(h)


def hotpath_dev_sig(lib, idx, opt, n_pairs, max_len, d_seq, d_off, d_pair_batch, d_sig, n_batches=1, id0=0):
    """hotpath_dev that also writes the pair signatures (device pointer d_sig: n_pairs x 3 uint64) for dist.global_markdup."""
    summary = np.zeros(8, dtype=np.uint64)
    lib._chk(lib.l.ssg_hotpath_dev_sig(idx, _ptr(opt), C.c_int(n_pairs), C.c_int(max_len), C.c_void_p(d_seq), C.c_void_p(d_off), C.c_void_p(d_pair_batch),
                                       C.c_int(n_batches), C.c_int64(id0), _ptr(summary), None, C.c_void_p(d_sig)))
    return summary, None


def prof_get(lib, cap=64):
    names = (C.c_char_p * cap)()
    ms = (C.c_double * cap)()
    cnt = (C.c_long * cap)()
    n = lib.l.ssg_prof_get(C.c_int(cap), names, ms, cnt)
    out = {}
    for i in range(min(n, cap)):   # the instance for reads up to 255 bases keeps the kernel's plain name; the other one says so
        k = names[i].decode().replace("<false>", "").replace("<true>", "<wide>").strip("()")
        a, b = out.get(k, (0.0, 0))
        out[k] = (a + ms[i], b + cnt[i])
    return out


def sam_format(lib, idx, opt, res, names, seq, off, quals=None, rg_id=""):
    n = 2 * res.n_pairs
    NA = (C.c_char_p * n)(*[s.encode() for s in names])
    QA = (C.c_char_p * n)(*[s.encode() for s in quals]) if quals is not None else None
    sam = C.c_char_p()
    sam_off = np.zeros(n + 1, dtype=np.int64)
    lib._chk(lib.l.ssg_sam_format(idx, _ptr(opt), res.h, C.c_int(res.n_pairs), NA, _ptr(seq), _ptr(off), QA, None, rg_id.encode(), C.byref(sam), _ptr(sam_off)))
    text = C.string_at(sam, int(sam_off[n])).decode()
    lib.l.ssg_free(sam)
    return text, sam_off


def bam_format(lib, idx, opt, res, names, seq, off, quals=None, rg_id=""):
    """ssg_bam_format: the host formatter's BAM records of a PeResult (what `sambamba view -S -f bam` makes of ssg_sam_format's lines)."""
    n = 2 * res.n_pairs
    NA = (C.c_char_p * n)(*[s.encode() for s in names])
    QA = (C.c_char_p * n)(*[(s.encode() if s is not None else None) for s in quals]) if quals is not None else None
    bam = C.c_void_p()
    bam_off = np.zeros(n + 1, dtype=np.int64)
    lib._chk(lib.l.ssg_bam_format(idx, _ptr(opt), res.h, C.c_int(res.n_pairs), NA, _ptr(seq), _ptr(off), QA, rg_id.encode(), C.byref(bam), _ptr(bam_off)))
    data = C.string_at(bam, int(bam_off[n]))
    lib.l.ssg_free(bam)
    return data, bam_off


BAM_CAND_DT = np.dtype([("pair", "i8"), ("first_rec", "i8"), ("n_rec", "i8"), ("byte_off", "i8"), ("n_bytes", "i8")])


def _take_pe_bam(lib, h, n_batches):
    l = lib.l
    for f in ("ssg_pe_bam_data", "ssg_pe_bam_cands", "ssg_pe_bam_pes", "ssg_pe_bam_stats"):
        getattr(l, f).restype = C.c_void_p
        getattr(l, f).argtypes = [C.c_void_p]
    for f in ("ssg_pe_bam_bytes", "ssg_pe_bam_n_rec", "ssg_pe_bam_n_cand"):
        getattr(l, f).restype = C.c_int64
        getattr(l, f).argtypes = [C.c_void_p]
    l.ssg_pe_bam_free.argtypes = [C.c_void_p]
    nb, nc = l.ssg_pe_bam_bytes(h), l.ssg_pe_bam_n_cand(h)
    out = {"bam": C.string_at(l.ssg_pe_bam_data(h), nb) if nb else b"", "n_rec": l.ssg_pe_bam_n_rec(h),
           "cands": np.frombuffer((C.c_char * (nc * BAM_CAND_DT.itemsize)).from_address(l.ssg_pe_bam_cands(h)), dtype=BAM_CAND_DT, count=nc).copy() if nc else np.zeros(0, BAM_CAND_DT),
           "pes": np.frombuffer((C.c_char * (n_batches * 4 * PESTAT_DT.itemsize)).from_address(l.ssg_pe_bam_pes(h)), dtype=PESTAT_DT, count=n_batches * 4).copy(),
           "stats": np.ctypeslib.as_array(C.cast(l.ssg_pe_bam_stats(h), C.POINTER(C.c_uint64)), shape=(8,)).copy()}
    l.ssg_pe_bam_free(h)
    return out


def mem_process_pairs_bam(lib, idx, opt, seq, off, names, quals=None, pair_batch=None, n_batches=1, id0=0, pes0=None, rg_id=""):
    """ssg_mem_process_pairs_bam: parsed reads in, BAM record bytes (made on the device) out."""
    n_pairs = (len(off) - 1) // 2
    n = 2 * n_pairs
    pair_batch = np.ascontiguousarray(pair_batch if pair_batch is not None else np.zeros(n_pairs, dtype=np.int32), dtype=np.int32)
    NA = (C.c_char_p * n)(*[s.encode() for s in names])
    QA = (C.c_char_p * n)(*[(s.encode() if s is not None else None) for s in quals]) if quals is not None else None
    h = C.c_void_p()
    p0 = _ptr(pes0) if pes0 is not None else None
    lib._chk(lib.l.ssg_mem_process_pairs_bam(idx, _ptr(opt), C.c_int(n_pairs), _ptr(seq), _ptr(off), NA, QA, _ptr(pair_batch), C.c_int(n_batches), C.c_int64(id0), p0,
                                             rg_id.encode() if rg_id else None, C.byref(h)))
    return _take_pe_bam(lib, h, n_batches)


def mem_process_fastq_bam(lib, idx, opt, text, rec_off, pair_batch=None, n_batches=1, id0=0, pes0=None, rg_id=""):
    """ssg_mem_process_fastq_bam: FASTQ text (bytes) of plain four-line records + the offset of every record's '@' in read order."""
    rec_off = np.ascontiguousarray(rec_off, dtype=np.int64)
    n_pairs = len(rec_off) // 2
    pair_batch = np.ascontiguousarray(pair_batch if pair_batch is not None else np.zeros(n_pairs, dtype=np.int32), dtype=np.int32)
    texts = [text] if isinstance(text, (bytes, bytearray)) else list(text)          # one piece, or several (the input files of a batch)
    bufs = [np.frombuffer(t, dtype=np.uint8) for t in texts]
    PA = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
    nbytes = np.array([len(t) for t in texts], dtype=np.int64)
    h = C.c_void_p()
    p0 = _ptr(pes0) if pes0 is not None else None
    lib._chk(lib.l.ssg_mem_process_fastq_bam(idx, _ptr(opt), C.c_int(n_pairs), PA, _ptr(nbytes), C.c_int(len(bufs)), _ptr(rec_off), _ptr(pair_batch), C.c_int(n_batches),
                                             C.c_int64(id0), p0, rg_id.encode() if rg_id else None, C.byref(h)))
    return _take_pe_bam(lib, h, n_batches)
