"""Shared test helpers: seeded inputs and the comparison routines used by both the CPU-side tests
(host-emulation build of the kernel sources) and the `-m gpu` parity tests (HIP build)."""
import os

import numpy as np

import simreads
from speedseq_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
EXAMPLE_FA = os.path.join(GOLDEN, "chr20_slice.fa")

KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000


def mutate(rng, q, max_sub=8, max_indel=3):
    tl = list(q)
    for _ in range(int(rng.integers(0, max_sub))):
        if not tl:
            break
        p = int(rng.integers(0, len(tl)))
        tl[p] = (tl[p] + 1) % 4
    for _ in range(int(rng.integers(0, max_indel))):
        if len(tl) < 2:
            break
        p = int(rng.integers(0, len(tl)))
        if rng.random() < 0.5:
            del tl[p:p + int(rng.integers(1, 8))]
        else:
            tl[p:p] = list(rng.integers(0, 4, size=int(rng.integers(1, 8))))
    return tl


def make_extend_jobs(n, seed, max_qlen=200):
    rng = np.random.default_rng(seed)
    jobs, qs, ts, qo, to = [], [], [], 0, 0
    for _ in range(n):
        qlen = int(rng.integers(1, max_qlen))
        q = rng.integers(0, 4, size=qlen, dtype=np.uint8)
        mode = rng.integers(0, 4)
        tl = list(q)
        if mode >= 1:
            tl = mutate(rng, q, 6, 1)
        if mode >= 2:
            tl = mutate(rng, np.array(tl, dtype=np.uint8), 3, 3)
        if mode == 3 and len(tl) > 10:
            tl = tl[:int(len(tl) * rng.random()) + 1] + list(rng.integers(0, 4, size=30))
        tl += list(rng.integers(0, 4, size=int(rng.integers(0, 120))))
        t = np.array(tl, dtype=np.uint8)
        if rng.random() < 0.1:
            q[rng.integers(0, qlen)] = 4
        h0 = int(rng.integers(1, 160))
        w = int(rng.choice([100, 200, 5, 20]))
        zd = int(rng.choice([100, 100, 0, 20]))
        jobs.append((qo, qlen, to, len(t), w, 5, zd, h0))
        qs.append(q)
        ts.append(t)
        qo += qlen
        to += len(t)
    return np.array(jobs, dtype=capi.EXT_JOB_DT), qs, ts


def make_local_jobs(n, seed, qlens=(150, 150, 100, 76, 36, 250, 200)):
    rng = np.random.default_rng(seed)
    jobs, qs, ts, qo, to = [], [], [], 0, 0
    for _ in range(n):
        qlen = int(rng.choice(list(qlens)))
        q = rng.integers(0, 4, size=qlen, dtype=np.uint8)
        mid = mutate(rng, q) if rng.random() < 0.8 else list(rng.integers(0, 4, size=50))
        if rng.random() < 0.3:
            mid = mid[:len(mid) // 2]
        t = np.array(list(rng.integers(0, 4, size=int(rng.integers(0, 300)))) + mid +
                     list(rng.integers(0, 4, size=int(rng.integers(0, 300)))), dtype=np.uint8)
        if rng.random() < 0.3:
            t = np.concatenate([t, np.array(mutate(rng, q), dtype=np.uint8)])
        xtra = KSW_XSUBO | KSW_XSTART | (KSW_XBYTE if qlen < 250 else 0) | 19
        jobs.append((qo, qlen, to, len(t), xtra, 0))
        qs.append(q)
        ts.append(t)
        qo += qlen
        to += len(t)
    return np.array(jobs, dtype=capi.SW_JOB_DT), qs, ts


def make_global_jobs(n, seed):
    rng = np.random.default_rng(seed)
    jobs, qs, ts, qo, to = [], [], [], 0, 0
    for _ in range(n):
        qlen = int(rng.integers(5, 250))
        q = rng.integers(0, 4, size=qlen, dtype=np.uint8)
        t = np.array(mutate(rng, q), dtype=np.uint8)
        w = abs(len(t) - qlen) + int(rng.integers(3, 40))
        jobs.append((qo, qlen, to, len(t), w, 0))
        qs.append(q)
        ts.append(t)
        qo += qlen
        to += len(t)
    return np.array(jobs, dtype=capi.GLB_JOB_DT), qs, ts


def sim_reads(n_pairs, seed, read_len=150, fasta=EXAMPLE_FA, **kw):
    contigs = simreads.read_fasta(fasta)
    pairs = simreads.simulate(contigs, n_pairs, seed=seed, read_len=read_len, **kw)
    seqs = []
    for _, r1, r2 in pairs:
        seqs += [r1, r2]
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return pairs, seqs, np.concatenate(seqs), off


REG_FIELDS = ["rb", "re", "qb", "qe", "rid", "score", "truesc", "sub", "csub", "w", "seedcov", "seedlen0", "n_comp", "frac_rep"]


def check_extend(lib, oracle, n, seed, max_qlen=200):
    jobs, qs, ts = make_extend_jobs(n, seed, max_qlen=max_qlen)
    res, cells = lib.extend_batch(lib.opt_init(), jobs, np.concatenate(qs), np.concatenate(ts))
    for i in range(n):
        o = oracle.extend2(qs[i], ts[i], int(jobs[i]["w"]), 5, int(jobs[i]["zdrop"]), int(jobs[i]["h0"]))
        assert o == tuple(int(x) for x in res[i]), (i, jobs[i], o, res[i])
    assert cells > 0


def check_extend_lane(lib, oracle, n, seed, workdir, qcaps=(72, 136, 256)):
    """ksw_extend2 through the lane-per-extension kernel code of the product's mem_chain2aln path (ssg_k_ext_lane's ln_extend2: packed 13-bit
    cells in LDS, 6-bit score table, targets read from the 2-bit reference) against the oracle: the jobs of make_extend_jobs (w 5 / 20 / 100 /
    200, zdrop 0 / 20 / 100, N in the query, h0 1..160) plus h0 up to the 13-bit ceiling, their targets laid out as a reference of their
    own, read forward, backward and from the reverse strand (the four ways the product's left / right extensions walk the reference)."""
    rng = np.random.default_rng(seed)
    fa = os.path.join(str(workdir), "extlane_%d.fa" % seed)
    done = 0
    for qcap in qcaps:
        jobs, qs, ts = make_extend_jobs(n, seed + qcap, max_qlen=min(qcap, 318) + 1)
        for i in range(n):   # a share of the jobs with start scores near the ceiling of the packed cells (h0 + qlen * a + end_bonus < 8191)
            if rng.random() < 0.15:
                jobs[i]["h0"] = 8190 - int(jobs[i]["qlen"]) - 5 - int(rng.integers(0, 60))
        tcat = np.concatenate(ts)
        with open(fa, "w") as f:
            f.write(">t\n")
            txt = "".join("ACGT"[c] for c in tcat)
            for k in range(0, len(txt), 80):
                f.write(txt[k:k + 80] + "\n")
        idx = lib.index_build_fasta(fa)
        L = int(tcat.size)
        starts = jobs["toff"].astype(np.int64)
        tl = jobs["tlen"].astype(np.int64)
        comp = lambda a: (3 - a).astype(np.uint8)
        modes = [("fwd +1", starts, 1, lambda t: t), ("fwd -1", starts + tl - 1, -1, lambda t: t[::-1]),
                 ("rev +1", 2 * L - starts - tl, 1, lambda t: comp(t[::-1])), ("rev -1", 2 * L - 1 - starts, -1, lambda t: comp(t))]
        for name, tpos, d, view in modes:
            res, cells = lib.extend_lane_batch(idx, lib.opt_init(), jobs, tpos, d, np.concatenate(qs), qcap)
            assert cells > 0
            for i in range(n):
                o = oracle.extend2(qs[i], np.ascontiguousarray(view(ts[i])), int(jobs[i]["w"]), 5, int(jobs[i]["zdrop"]), int(jobs[i]["h0"]))
                assert o == tuple(int(x) for x in res[i]), (qcap, name, i, jobs[i], o, res[i])
            done += n
        lib.index_destroy(idx)
    return done


def check_local_lane(lib, oracle, n, seed, workdir, lanes=(1, 2, 4), scores=None, qlens=None):
    """ksw_align2 as mate rescue runs it in the product path (k_mswlane.h: forward pass by the lane kernel -- strips of 8 target rows in
    registers, packed 13-bit strip boundaries in LDS, 5-bit score table, targets from the 2-bit reference --, reverse pass by the wave code)
    against the oracle: make_local_jobs' queries (36 .. 250 bases, some with N) and targets (hits, half hits, second hits, none), the targets
    laid out as a reference of their own and read from both strands, 1 / 2 / 4 lanes per job."""
    rng = np.random.default_rng(seed)
    jobs, qs, ts = make_local_jobs(n, seed, qlens) if qlens else make_local_jobs(n, seed)
    for i in range(n):
        if rng.random() < 0.2:   # N bases in the query
            q = qs[i].copy(); q[rng.random(q.size) < 0.03] = 4; qs[i] = q
        if ts[i].size == 0:
            ts[i] = rng.integers(0, 4, size=40, dtype=np.uint8)
    to = 0
    for i in range(n):
        jobs[i]["toff"] = to; jobs[i]["tlen"] = ts[i].size; to += ts[i].size
    tcat = np.concatenate(ts)
    fa = os.path.join(str(workdir), "locallane_%d.fa" % seed)
    with open(fa, "w") as f:
        f.write(">t\n")
        txt = "".join("ACGT"[c] for c in tcat)
        for k in range(0, len(txt), 80):
            f.write(txt[k:k + 80] + "\n")
    idx = lib.index_build_fasta(fa)
    L = int(tcat.size)
    starts = jobs["toff"].astype(np.int64)
    tl = jobs["tlen"].astype(np.int64)
    opt, oopt = lib.opt_init(), None
    if scores:   # (a, b, o_del, e_del, o_ins, e_ins)
        for k, v in zip(("a", "b", "o_del", "e_del", "o_ins", "e_ins"), scores):
            opt[k] = v
        m = opt["mat"][0]
        for x in range(4):
            for y in range(4):
                m[x * 5 + y] = scores[0] if x == y else -scores[1]
        oopt = oracle.opt_scores(*scores)
    done = taken = 0
    rc = lambda a: np.where(a[::-1] < 4, 3 - a[::-1], 4).astype(np.uint8)
    for name, tpos, view, qview in (("fwd", starts, lambda t: t, lambda q: q), ("rev", 2 * L - starts - tl, rc, rc)):   # the reverse strand reads the windows reverse-complemented: so are the queries, or nothing would hit
        qv = [np.ascontiguousarray(qview(q)) for q in qs]
        want = [oracle.align2(qv[i], np.ascontiguousarray(view(ts[i])), int(jobs[i]["xtra"]), oopt) for i in range(n)]
        for nl in lanes:
            res, from_lane = lib.align2_lane_batch(idx, opt, jobs, tpos, np.concatenate(qv), nl)
            for i in range(n):
                assert want[i] == tuple(int(x) for x in res[i]), (nl, name, i, jobs[i], want[i], res[i], int(from_lane[i]))
            done += n
            taken += int((from_lane > 0).sum())
            assert (from_lane == 2).sum() > n // 4, "the reverse passes did not go through the lane kernel"
    lib.index_destroy(idx)
    return done, taken


def check_seeds(lib, oracle, n_pairs, seed, prefix=EXAMPLE_FA, read_len=150):
    """the seeds mem_chain visits (interval -> sampled occurrences -> bwt_sa + bns_intv2rid, upstream's order) straight from ssg_k_sal against the oracle"""
    oidx, gidx = oracle.idx_load(prefix), lib.index_load(prefix)
    _, seqs, seq, off = sim_reads(n_pairs, seed, read_len, fasta=prefix)
    seed_off, seeds, rids = lib.seeds_batch(gidx, lib.opt_init(), seq, off)
    tot = 0
    for r, s in enumerate(seqs):
        o = oracle.seeds(oidx, s)
        g = seeds[seed_off[r]:seed_off[r + 1]]
        assert len(o) == len(g), (r, len(o), len(g))
        assert np.array_equal(o[:, 0], g["rbeg"]) and np.array_equal(o[:, 1], g["qbeg"]) and np.array_equal(o[:, 2], g["len"]), r
        assert np.array_equal(o[:, 3], rids[seed_off[r]:seed_off[r + 1]]), r
        tot += len(o)
    lib.index_destroy(gidx)
    return tot


def check_local(lib, oracle, n, seed):
    jobs, qs, ts = make_local_jobs(n, seed)
    res = lib.align2_batch(lib.opt_init(), jobs, np.concatenate(qs), np.concatenate(ts))
    for i in range(n):
        o = oracle.align2(qs[i], ts[i], int(jobs[i]["xtra"]))
        assert o == tuple(int(x) for x in res[i]), (i, jobs[i], o, res[i])


def check_global(lib, oracle, n, seed):
    jobs, qs, ts = make_global_jobs(n, seed)
    sc, nc, cg = lib.global_batch(lib.opt_init(), jobs, np.concatenate(qs), np.concatenate(ts))
    for i in range(n):
        osc, on, ocg = oracle.global2(qs[i], ts[i], int(jobs[i]["w"]))
        assert osc == sc[i] and on == nc[i] and np.array_equal(ocg[:on], cg[i, :on]), (i, jobs[i])


def check_smem(lib, oracle, n_pairs, seed, read_len=150, prefix=EXAMPLE_FA, n_frac=0.0, cap=96):
    """seeding intervals (upstream mem_collect_intv) of every read against the oracle; n_frac: share of the bases turned into N"""
    oidx, gidx = oracle.idx_load(prefix), lib.index_load(prefix)
    _, seqs, seq, off = sim_reads(n_pairs, seed, read_len, fasta=prefix)
    if n_frac > 0:
        rng = np.random.RandomState(seed)
        seq = seq.copy()
        seq[rng.random_sample(seq.size) < n_frac] = 4
        seqs = [seq[off[i]:off[i + 1]] for i in range(len(off) - 1)]
    intv, cnt = lib.smem_batch(gidx, lib.opt_init(), seq, off, cap=cap)
    for r, s in enumerate(seqs):
        o = oracle.collect_intv(oidx, s)
        assert len(o) == cnt[r] and np.array_equal(o, intv[r, :cnt[r]]), r
    lib.index_destroy(gidx)


def check_pe_sam(lib, oracle, n_pairs, seed, read_len=150, n_threads=8, prefix=EXAMPLE_FA, **kw):
    """Whole `bwa mem` PE hot path: SAM text from the device records must equal the oracle's."""
    oidx, gidx = oracle.idx_load(prefix), lib.index_load(prefix)
    pairs, seqs, seq, off = sim_reads(n_pairs, seed, read_len, fasta=prefix, **kw)
    names = []
    for nm, _, _ in pairs:
        names += [nm, nm]
    quals = ["I" * len(s) for s in seqs]
    opt = lib.opt_init()
    res = capi.mem_process_pairs(lib, gidx, opt, seq, off, id0=0)
    text, _ = capi.sam_format(lib, gidx, opt, res, names, seq, off, quals, "grp1")
    otext, _, opes = oracle.process_pairs(oidx, seq, off, names, quals, 0, "grp1", n_threads)
    for f in ("low", "high", "failed", "avg", "std"):
        assert np.array_equal(res.pes[f][:4], opes[f]), (f, res.pes, opes)
    if text != otext:
        a, b = text.split("\n"), otext.split("\n")
        assert len(a) == len(b), (len(a), len(b))
        for x, y in zip(a, b):
            assert x == y, "\n%s\n%s" % (x, y)
    stats = res.stats.copy()
    res.close()
    lib.index_destroy(gidx)
    return text, stats


def check_pe_edge_cases(lib, oracle):
    """Ragged / degenerate inputs the reference's aligner meets in practice: reads of different lengths,
    a read below min_seed_len, all-N reads, a random (unmappable) mate, exact duplicates, a pair count of 1."""
    prefix = EXAMPLE_FA
    oidx, gidx = oracle.idx_load(prefix), lib.index_load(prefix)
    contigs = simreads.read_fasta(prefix)
    ref = contigs[0][1]
    rng = np.random.default_rng(77)
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)

    def frag(pos, l):
        return ref[pos:pos + l].copy()

    def rc(s):
        return comp[s[::-1]]

    pairs = []
    pairs.append((frag(1000, 150), rc(frag(1300, 150))))                       # plain proper pair
    pairs.append((frag(1000, 150), rc(frag(1300, 150))))                       # exact duplicate of it
    pairs.append((frag(5000, 100), rc(frag(5250, 76))))                        # ragged lengths
    pairs.append((frag(9000, 12), rc(frag(9300, 150))))                        # read shorter than min_seed_len
    pairs.append((np.full(150, 4, dtype=np.uint8), rc(frag(12000, 150))))      # all-N read 1
    pairs.append((np.full(80, 4, dtype=np.uint8), np.full(80, 4, dtype=np.uint8)))  # both all-N
    pairs.append((frag(20000, 150), rng.integers(0, 4, size=150, dtype=np.uint8)))  # random mate -> rescue attempt
    pairs.append((rng.integers(0, 4, size=150, dtype=np.uint8), rng.integers(0, 4, size=150, dtype=np.uint8)))  # both random
    a = frag(30000, 150)
    a[75] = 4
    pairs.append((a, rc(frag(30310, 150))))                                    # N in the middle
    pairs.append((np.concatenate([frag(40000, 80), frag(90000, 70)]), rc(frag(40300, 150))))  # chimeric read 1
    for k in range(40):                                                         # enough proper pairs for the insert-size model
        p = 50000 + 997 * k
        pairs.append((frag(p, 150), rc(frag(p + 250 + (k % 7) * 10, 150))))
    seqs = []
    for r1, r2 in pairs:
        seqs += [r1, r2]
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    seq = np.concatenate(seqs)
    names = []
    for i in range(len(pairs)):
        names += ["e%d" % i, "e%d" % i]
    quals = ["5" * len(s) for s in seqs]
    opt = lib.opt_init()
    res = capi.mem_process_pairs(lib, gidx, opt, seq, off, id0=7)
    text, _ = capi.sam_format(lib, gidx, opt, res, names, seq, off, quals, "g")
    otext, _, _ = oracle.process_pairs(oidx, seq, off, names, quals, 14, "g", 2)
    assert text == otext, "\n".join(x + "\n" + y for x, y in zip(text.split("\n"), otext.split("\n")) if x != y)
    res.close()
    # a single pair: insert-size inference fails for every orientation, pairing falls back
    one_seq, one_off = np.concatenate(seqs[:2]), np.array([0, 150, 300], dtype=np.int64)
    res = capi.mem_process_pairs(lib, gidx, opt, one_seq, one_off, id0=0)
    t1, _ = capi.sam_format(lib, gidx, opt, res, names[:2], one_seq, one_off, quals[:2], "")
    o1, _, opes = oracle.process_pairs(oidx, one_seq, one_off, names[:2], quals[:2], 0, "", 1)
    assert t1 == o1 and all(int(x) == 1 for x in res.pes["failed"][:4])
    res.close()
    lib.index_destroy(gidx)
    return text


def sam_primary_ends(text, contig_names):
    """Per pair: the two primary records as samblaster sees them (capi.SBL_END_DT), from SAM text."""
    import re
    idx = {n: i for i, n in enumerate(contig_names)}
    ends, cur, name = [], {}, None

    def flush():
        if cur:
            ends.append(cur.get(0x40, (-1, 0, 0x4, 0, 0, 0)))
            ends.append(cur.get(0x80, (-1, 0, 0x4, 0, 0, 0)))

    for line in text.split("\n"):
        if not line or line[0] == "@":
            continue
        f = line.split("\t")
        if f[0] != name:
            flush()
            cur, name = {}, f[0]
        flag = int(f[1])
        if flag & 0x900:
            continue
        ops = re.findall(r"(\d+)([MIDNSH=X])", f[5])
        lclip = rclip = ralen = 0
        first = True
        for n, op in ops:
            n = int(n)
            if op in "SH":
                if first:
                    lclip += n
                rclip += n
            else:
                first = False
                rclip = 0
                if op in "MDN=X":
                    ralen += n
        seq = -1 if (flag & 4) or f[2] == "*" else idx[f[2]]
        cur[flag & 0xc0] = (seq, int(f[3]), flag, lclip, rclip if ops else 0, ralen)
    flush()
    return np.array(ends, dtype=capi.SBL_END_DT)


def oracle_dup_flags(oracle, sam_text, header):
    """Run the oracle samblaster over SAM text; returns per-pair dup flags (from read1 primaries)."""
    out = oracle.samblaster(header + sam_text)
    flags, name = [], None
    for line in out.split("\n"):
        if not line or line[0] == "@":
            continue
        f = line.split("\t")
        if f[0] != name:
            name = f[0]
            flags.append(1 if int(f[1]) & 0x400 else 0)
    return np.array(flags, dtype=np.uint8), out


def check_dedup(lib, oracle, n_pairs, seed, dup_frac=0.2):
    prefix = EXAMPLE_FA
    oidx = oracle.idx_load(prefix)
    pairs, seqs, seq, off = sim_reads(n_pairs, seed, dup_frac=dup_frac)
    names = []
    for i, (nm, _, _) in enumerate(pairs):
        names += [nm, nm]
    otext, _, _ = oracle.process_pairs(oidx, seq, off, names, None, 0, "", 8)
    header = "@SQ\tSN:20_slice\tLN:321635\n"
    oflags, _ = oracle_dup_flags(oracle, otext, header)
    ends = sam_primary_ends(otext, ["20_slice"])
    dup = capi.sbl_markdup(lib, ends)
    assert len(dup) == len(oflags) == n_pairs
    assert np.array_equal(dup, oflags), (int(dup.sum()), int(oflags.sum()))
    assert dup.sum() > 0
    return int(dup.sum())


def check_align1(lib, oracle, n_pairs, seed, read_len=150, prefix=EXAMPLE_FA, chain_opt=None):
    """chain_opt: (drop_ratio, mask_level, min_chain_weight, max_chain_extend, max_chain_gap) for the chain filter, both sides"""
    oidx, gidx = oracle.idx_load(prefix), lib.index_load(prefix)
    _, seqs, seq, off = sim_reads(n_pairs, seed, read_len, fasta=prefix)
    opt, oopt = lib.opt_init(), None
    if chain_opt is not None:
        for f, v in zip(("drop_ratio", "mask_level", "min_chain_weight", "max_chain_extend", "max_chain_gap"), chain_opt):
            opt[f] = v
        oopt = oracle.opt_chain(*chain_opt)
    ro, regs, st = lib.align1_batch(gidx, opt, seq, off)
    oro, oregs = oracle.align1_batch(oidx, seq, off, oopt)
    assert np.array_equal(ro, oro)
    for f in REG_FIELDS:
        assert np.array_equal(regs[f], oregs[f]), f
    lib.index_destroy(gidx)
    return len(regs)



def reads_with_inner_repeats(fasta, n, seed, rl=250):
    """reads that carry the same reference segment two or three times, further apart than the band: seeds at EQUAL reference positions that do not merge
    (a second, then a third chain at one position: upstream's order among equal positions, the give-up of the wave kernels' ranked form)"""
    rng = np.random.default_rng(seed)
    ctg = simreads.read_fasta(fasta)
    ref = np.asarray(ctg[0][1] if isinstance(ctg[0], tuple) else ctg[0], dtype=np.uint8)
    out = []
    for _ in range(n):
        copies = int(rng.choice([2, 3, 3]))
        xl = int(rng.integers(25, 45))
        p = int(rng.integers(1000, len(ref) - 1000))
        x = ref[p:p + xl]
        gap = int(rng.integers(105, 125)) if copies == 2 else max(5, (rl - copies * xl) // (copies - 1) - 1)
        parts = []
        for c in range(copies):
            parts.append(x)
            if c + 1 < copies:
                parts.append(rng.integers(0, 4, size=gap).astype(np.uint8))
        s = np.concatenate(parts)[:rl]
        if len(s) < rl:
            s = np.concatenate([s, rng.integers(0, 4, size=rl - len(s)).astype(np.uint8)])
        if rng.random() < 0.5:
            s = (3 - s[::-1]).astype(np.uint8)
        out.append(s)
    return out


def check_align1_reads(lib, oracle, seqs, prefix=EXAMPLE_FA):
    """mem_align1_core on given reads: every region field against the oracle"""
    oidx, gidx = oracle.idx_load(prefix), lib.index_load(prefix)
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    seq = np.concatenate(seqs)
    ro, regs, st = lib.align1_batch(gidx, lib.opt_init(), seq, off)
    oro, oregs = oracle.align1_batch(oidx, seq, off)
    assert np.array_equal(ro, oro)
    for f in REG_FIELDS:
        assert np.array_equal(regs[f], oregs[f]), f
    lib.index_destroy(gidx)
    return len(regs)

def repeat_reference(oracle, dirname, seed=5, n_copies=(300, 70), fam_len=(600, 350), unique=40000, max_div=0.03):
    """Writes a small repeat-rich reference (two planted families at 0-3 % divergence, both strands)
    and its index (built by the oracle) into dirname; returns the prefix.  Reads drawn from it carry
    hundreds to thousands of seeds -- the regime of the wave-per-read chaining kernels."""
    prefix = os.path.join(str(dirname), "repeats.fa")
    if os.path.exists(prefix + ".bwt"):
        return prefix
    rng = np.random.default_rng(seed)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    fams = [rng.integers(0, 4, size=l).astype(np.uint8) for l in fam_len]
    pieces = []
    for f, n in zip(fams, n_copies):
        for _ in range(n):
            c = f.copy()
            m = rng.random(c.size) < rng.random() * max_div
            c[m] = rng.integers(0, 4, size=int(m.sum()))
            if rng.random() < 0.5:
                c = comp[c[::-1]]
            pieces.append(c)
    n_sp = len(pieces) + 1
    spacers = [rng.integers(0, 4, size=max(20, int(unique / n_sp))).astype(np.uint8) for _ in range(n_sp)]
    order = rng.permutation(len(pieces))
    seq = [spacers[0]]
    for k, i in enumerate(order):
        seq += [pieces[i], spacers[k + 1]]
    seq = np.concatenate(seq)
    cut = seq.size * 2 // 3
    with open(prefix, "w") as fh:
        for name, s in (("rep1", seq[:cut]), ("rep2", seq[cut:])):
            fh.write(">%s\n" % name)
            txt = "".join("ACGT"[x] for x in s)
            for i in range(0, len(txt), 60):
                fh.write(txt[i:i + 60] + "\n")
    oracle.idx_build(prefix, save=True)
    return prefix


def sbl_streams_from_bits(text, bits, mate, exclude_dups=True, add_mate_tags=True):
    """The three streams samblaster's writer makes of name-grouped SAM lines and the per-line decisions of the device (SSG_SBL_* bits,
    mate line): 0x400 into FLAG of a duplicate block's lines, MC / MQ from the mate's primary line, both primaries of a discordant pair
    (read 1 first) to the discordant stream, splitter lines with _1 / _2 to the splitter stream -- the emit rules of
    speedseq_amd/host/samblaster_main.cpp, restated here so that a device step can be compared with the oracle's streams line by line."""
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    assert len(lines) == len(bits) == len(mate), (len(lines), len(bits), len(mate))
    f = [l.split("\t", 6) for l in lines]
    main, disc, spl = [], [], []

    def emit(i, flag, patch, suffix):
        l = lines[i]
        if patch or suffix:
            x = f[i]
            l = x[0] + (suffix or "") + "\t" + str(flag) + l[len(x[0]) + 1 + len(x[1]):]
        m = int(mate[i])
        if add_mate_tags and m >= 0:
            if "\tMC:Z:" not in l:
                l += "\tMC:Z:" + f[m][5]
            if "\tMQ:i:" not in l:
                l += "\tMQ:i:" + f[m][4]
        return l

    n, b0 = len(lines), 0
    while b0 < n:
        b1 = b0 + 1
        while b1 < n and f[b1][0] == f[b0][0]:
            b1 += 1
        dup = any(bits[i] & 1 for i in range(b0, b1))
        d1 = d2 = -1
        for i in range(b0, b1):
            flag = int(f[i][1])
            if bits[i] & 2:
                if flag & 0x40:
                    d1 = i
                else:
                    d2 = i
            main.append(emit(i, flag | (0x400 if bits[i] & 1 else 0), bool(bits[i] & 1), None))
        if d1 >= 0 and d2 >= 0:
            for i in (d1, d2):
                disc.append(emit(i, int(f[i][1]) | (0x400 if dup else 0), dup, None))
        for i in range(b0, b1):
            if bits[i] & 4:
                flag = int(f[i][1])
                spl.append(emit(i, flag | (0x400 if dup else 0), True, "_1" if flag & 0x40 else "_2"))
        b0 = b1
    return main, disc, spl


def oracle_streams(oracle, text, header):
    """oracle samblaster (the reference's switches) over header + text: record lines of the three streams"""
    out = oracle.samblaster(header + text)
    rec = lambda t: [l for l in t.split("\n") if l and l[0] != "@"]
    return rec(out), rec(oracle.last_discordants), rec(oracle.last_splitters)


def check_hotpath(lib, oracle, n_pairs, per, read_len, to_dev):
    """ssg_hotpath_dev_ex (the bench's step: alignment, duplicate marking, discordant / splitter classification, all on the device) on
    device-resident reads in n_pairs / per upstream batches, against the oracle's `bwa mem` (one insert-size model per upstream batch)
    + samblaster over the whole input: the records left in HBM, printed, are the oracle's SAM text; their per-line decisions
    (duplicate / discordant / splitter bits, MC / MQ source line) give the oracle's three streams line for line.
    to_dev(numpy array) -> (keep-alive object, device pointer)."""
    from speedseq_amd import capi
    kw = dict(ins_mean=800, ins_std=150) if read_len > 200 else {}
    pairs, seqs, seq, off = sim_reads(n_pairs, seed=41, read_len=read_len, dup_frac=0.1, **kw)
    nb = n_pairs // per
    pb = (np.arange(n_pairs) // per).astype(np.int32)
    gidx, oidx = lib.index_load(EXAMPLE_FA), oracle.idx_load(EXAMPLE_FA)
    opt = lib.opt_init()
    keep = [to_dev(seq), to_dev(off), to_dev(pb)]
    (d_seq, d_off, d_pb) = [k[1] for k in keep]
    summary, dup = capi.hotpath_dev(lib, gidx, opt, n_pairs, read_len, d_seq, d_off, d_pb, nb, 0, True)
    s16, h = capi.hotpath_dev_ex(lib, gidx, opt, n_pairs, read_len, d_seq, d_off, d_pb, nb, 0, keep=True)
    names = []
    for nm, _, _ in pairs:
        names += [nm, nm]
    text = ""
    for b in range(nb):
        lo, hi = 2 * per * b, 2 * per * (b + 1)
        t, _, _ = oracle.process_pairs(oidx, seq[off[lo]:off[hi]], off[lo:hi + 1] - off[lo], names[lo:hi], None, lo, "", 4)
        text += t
    header = "@SQ\tSN:20_slice\tLN:321635\n"
    oflags, marked = oracle_dup_flags(oracle, text, header)
    assert np.array_equal(dup, oflags) and oflags.sum() > n_pairs // 40, (int(dup.sum()), int(oflags.sum()))
    assert int(s16[10]) == text.count("\n") and int(s16[1]) == int(oflags.sum()), (s16, text.count("\n"))
    res, bits, mate = capi.dev_records_download(lib, h, n_pairs)
    gtext, _ = capi.sam_format(lib, gidx, opt, res, names, seq, off, None, "")
    res.close(); capi.dev_records_free(lib, h)
    assert gtext == text                                        # the records the step left in HBM print as the oracle's SAM text
    gm, gd, gs = sbl_streams_from_bits(gtext, bits, mate)
    om, od, os_ = oracle_streams(oracle, text, header)
    assert gm == om and gd == od and gs == os_, (len(gm), len(om), len(gd), len(od), len(gs), len(os_))
    n_disc, n_spl = len(od) // 2, len(os_)
    assert (int(s16[8]), int(s16[9])) == (len(od), n_spl) and n_disc > 0 and (n_spl > 0 or n_pairs < 2000), (s16, n_disc, n_spl)
    return int(s16[10]), int(s16[1]), len(od), n_spl
