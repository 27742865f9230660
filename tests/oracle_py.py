"""ctypes binding of the CPU oracle (oracle/liboracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C

import numpy as np

from speedseq_amd.capi import ALNREG_DT, INTV_DT


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, so):
        self.l = C.CDLL(so)
        self.l.orc_idx_load.restype = C.c_void_p
        self.l.orc_idx_load.argtypes = [C.c_char_p]
        self.l.orc_idx_build_fasta.restype = C.c_void_p
        self.l.orc_idx_build_fasta.argtypes = [C.c_char_p]
        self.l.orc_idx_save.argtypes = [C.c_void_p, C.c_char_p]
        self.l.orc_idx_destroy.argtypes = [C.c_void_p]
        self.l.orc_api_opt_new.restype = C.c_void_p
        self.l.orc_api_align1_batch.restype = C.c_int64
        self.opt = C.c_void_p(self.l.orc_api_opt_new())

    def idx_load(self, prefix):
        h = self.l.orc_idx_load(prefix.encode())
        if not h:
            raise RuntimeError("oracle: cannot load index " + prefix)
        return C.c_void_p(h)

    def idx_build(self, fasta, save=True):
        h = self.l.orc_idx_build_fasta(fasta.encode())
        if not h:
            raise RuntimeError("oracle: cannot build index from " + fasta)
        h = C.c_void_p(h)
        if save:
            assert self.l.orc_idx_save(h, fasta.encode()) == 0
        return h

    def extend2(self, q, t, w, end_bonus, zdrop, h0):
        out = (C.c_int * 6)()
        self.l.orc_api_extend2(self.opt, C.c_int(len(q)), _ptr(q), C.c_int(len(t)), _ptr(t), C.c_int(w), C.c_int(end_bonus), C.c_int(zdrop), C.c_int(h0), out)
        return tuple(out)

    def opt_scores(self, a, b, o_del, e_del, o_ins, e_ins):
        """an option block of its own with this scoring (release: let it go)"""
        o = C.c_void_p(self.l.orc_api_opt_new())
        self.l.orc_api_opt_scores(o, C.c_int(a), C.c_int(b), C.c_int(o_del), C.c_int(e_del), C.c_int(o_ins), C.c_int(e_ins))
        return o

    def opt_chain(self, drop_ratio, mask_level, min_chain_weight, max_chain_extend, max_chain_gap):
        """an option block of its own with these chain-filter settings"""
        o = C.c_void_p(self.l.orc_api_opt_new())
        self.l.orc_api_opt_chain(o, C.c_float(drop_ratio), C.c_float(mask_level), C.c_int(min_chain_weight), C.c_int(max_chain_extend), C.c_int(max_chain_gap))
        return o

    def align2(self, q, t, xtra, opt=None):
        out = (C.c_int * 7)()
        self.l.orc_api_align2(opt or self.opt, C.c_int(len(q)), _ptr(q), C.c_int(len(t)), _ptr(t), C.c_int(xtra), out)
        return tuple(out)

    def global2(self, q, t, w, cap=64):
        n = C.c_int(0)
        cig = np.zeros(cap, dtype=np.uint32)
        sc = self.l.orc_api_global2(self.opt, C.c_int(len(q)), _ptr(q), C.c_int(len(t)), _ptr(t), C.c_int(w), C.byref(n), _ptr(cig), C.c_int(cap))
        return sc, n.value, cig

    def collect_intv(self, idx, seq, cap=4096):
        out = np.zeros(cap, dtype=INTV_DT)
        n = self.l.orc_api_collect_intv(self.opt, idx, C.c_int(len(seq)), _ptr(seq), _ptr(out), C.c_int(cap))
        return out[:n]

    def seeds(self, idx, seq, cap=1 << 16):
        """(rbeg, qbeg, len, rid) of every seed mem_chain visits for the read, in its order"""
        out = np.zeros((cap, 4), dtype=np.int64)
        self.l.orc_api_seeds.restype = C.c_int64
        n = self.l.orc_api_seeds(self.opt, idx, C.c_int(len(seq)), _ptr(seq), _ptr(out), C.c_int64(cap))
        assert n <= cap
        return out[:n]

    def align1_batch(self, idx, seq, off, opt=None):
        n = len(off) - 1
        reg_off = np.zeros(n + 1, dtype=np.int64)
        cap = 64 * n + 1024
        while True:
            out = np.zeros(cap, dtype=ALNREG_DT)
            tot = self.l.orc_api_align1_batch(opt or self.opt, idx, C.c_int(n), _ptr(seq), _ptr(off), _ptr(reg_off), _ptr(out), C.c_int64(cap))
            if tot <= cap:
                return reg_off, out[:tot]
            cap = tot

    def chain_exposure(self, idx, seq, off, n_threads=8, opt=None, per_read=False):
        """every read through both containers of mem_chain (orc_mem.c: the position-sorted array the kernels restate, and klib's B-tree as upstream uses it):
        dict(differ, gt9_and_dup, gt9, dup, reads) [+ per-read flags]"""
        n = len(off) - 1
        out = np.zeros(5, dtype=np.int64)
        which = np.zeros(n, dtype=np.int32) if per_read else None
        self.l.orc_api_chain_exposure(opt or self.opt, idx, C.c_int(n), _ptr(seq), _ptr(off), C.c_int(n_threads), _ptr(out), _ptr(which) if per_read else None)
        d = dict(differ=int(out[0]), gt9_and_dup=int(out[1]), gt9=int(out[2]), dup=int(out[3]), reads=int(out[4]))
        return (d, which) if per_read else d

    def lazyf(self, on=-1):
        """(calls, calls that differ) of orc_ksw_align2 under the lazy-F exposure count; on = 1 / 0 switches the counting on / off, -1 only reads"""
        out = np.zeros(2, dtype=np.uint64)
        self.l.orc_api_lazyf(C.c_int(on), _ptr(out))
        return int(out[0]), int(out[1])

    def samblaster(self, sam_text, exclude_dups=True, add_mate_tags=True, max_split=2, min_non_overlap=20):
        """Oracle samblaster over SAM text (via temp files); returns the marked SAM text."""
        import os
        import subprocess
        import tempfile
        exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "orc_bwa")
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "in.sam"), "w") as f:
                f.write(sam_text)
            cmd = [exe, "samblaster"] + (["--excludeDups"] if exclude_dups else []) + (["--addMateTags"] if add_mate_tags else []) + \
                  ["--maxSplitCount", str(max_split), "--minNonOverlap", str(min_non_overlap),
                   "--splitterFile", os.path.join(d, "spl.sam"), "--discordantFile", os.path.join(d, "disc.sam")]
            with open(os.path.join(d, "in.sam")) as fi:
                out = subprocess.run(cmd, stdin=fi, capture_output=True, text=True, check=True).stdout
            self.last_splitters = open(os.path.join(d, "spl.sam")).read()
            self.last_discordants = open(os.path.join(d, "disc.sam")).read()
        return out

    def process_pairs(self, idx, seq, off, names, quals=None, n_processed=0, rg_id="", n_threads=1, pes0=None):
        from speedseq_amd.capi import PESTAT_DT
        n = len(off) - 1
        NA = (C.c_char_p * n)(*[s.encode() for s in names])
        QA = (C.c_char_p * n)(*[s.encode() for s in quals]) if quals is not None else None
        pes = np.zeros(4, dtype=PESTAT_DT)
        sam_off = np.zeros(n + 1, dtype=np.int64)
        self.l.orc_api_process_pairs.restype = C.c_void_p
        p = self.l.orc_api_process_pairs(self.opt, idx, C.c_int(n // 2), _ptr(seq), _ptr(off), NA, QA, C.c_int64(n_processed), rg_id.encode(),
                                         C.c_int(n_threads), _ptr(pes0) if pes0 is not None else None, _ptr(pes), _ptr(sam_off))
        text = C.string_at(p, int(sam_off[n])).decode()
        self.l.orc_api_free(C.c_void_p(p))
        return text, sam_off, pes
