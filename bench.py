#!/usr/bin/env python3
"""bench.py -- `speedseq align` hot path (BWA-MEM seed/extend/pair + SAMBLASTER duplicate marking)
on MI355X.  Contract: python bench.py --gpus N --steps K --warmup W prints ONE JSON line on rank 0.

A "step" = one pass of the whole hot path (ssg_hotpath_dev) over one batch of synthetic 2x150 bp
pairs that is already resident in HBM: SMEM seeding -> SAL -> chaining -> banded SW extension ->
insert-size model -> mate rescue -> pairing/MAPQ -> CIGAR/NM/MD -> duplicate marking, records left
in HBM.  Reads shard across ranks with no data-path collective (weak scaling: every GPU aligns its
own pairs against its own index replica).

The reference FASTA the metric names (GRCh37) is not available offline; the workload uses a seeded
synthetic reference of GRCh37's size and contig-length proportions (order-6 Markov model of the
bundled chr20 slice + planted repeat families; stated in config.workload).  The FM-index
is built on the GPU (`bwa index` kernels, k_index.h) before the timed region.  After the timed region
a sample of the same batch is aligned by the CPU oracle and compared record by record with the GPU's
output (parity gate: a mismatch voids `value`); the same oracle run is the reported cpu_baseline.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

GRCH37 = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022, 141213431, 135534747, 135006516,
          133851895, 115169878, 107349540, 102531392, 90354753, 81195210, 78077248, 59128983, 63025520, 48129895, 51304566,
          155270560, 59373566, 16569]
GRCH37_NAMES = [str(i) for i in range(1, 23)] + ["X", "Y", "MT"]


_T0 = time.time()


def host_cpu_quota():
    """CPUs the container may use at once (cgroup v2 cpu.max / v1 cfs quota), None = no quota: the thread counts below are what was ASKED for"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            return None if q <= 0 else round(q / per, 2)
        except Exception:
            return None


def log(msg):
    """progress on stderr with the time since start: a run cut short still says where its time went"""
    sys.stderr.write("[bench %7.1f s] %s\n" % (time.time() - _T0, msg)); sys.stderr.flush()


def markov_table(order):
    """order-k transition CDF (4^k x 3 thresholds) trained on the reference's bundled chr20 slice (tests/golden/chr20_slice.fa,
    a copy of /root/reference/example/data/*.fasta), add-one smoothed"""
    import simreads
    codes = np.concatenate([c for _, c in simreads.read_fasta(os.path.join(ROOT, "tests", "golden", "chr20_slice.fa"))]).astype(np.int64)
    codes = codes[codes < 4]
    ctx = np.zeros(codes.size - order, dtype=np.int64)
    for k in range(order):
        ctx = ctx * 4 + codes[k:codes.size - order + k]
    cnt = np.ones((4 ** order, 4), dtype=np.float64)
    np.add.at(cnt, (ctx, codes[order:]), 1.0)
    cdf = np.cumsum(cnt / cnt.sum(1, keepdims=True), axis=1)
    return cdf[:, :3].astype(np.float32)


def synth_reference(total_len, seed, dev, order=6):
    """25 contigs with GRCh37's length proportions; sequence from a seeded order-k Markov model of the
    bundled chr20 slice; >= 5 % of the bases planted as repeat families of 10..10^4 copies at 0-10 % divergence, both strands."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lens = [max(2000, int(x * total_len / sum(GRCH37))) for x in GRCH37]
    L = sum(lens)
    cdf = torch.from_numpy(markov_table(order)).to(dev)
    S = 1 << 20 if L >= (1 << 26) else 1 << 12           # independent streams, each a contiguous stretch of the genome
    ref = torch.empty(L, dtype=torch.uint8, device=dev)
    syn = C.CDLL(os.path.join(ROOT, "tools", "synth", "libsynthref.so"))       # one kernel launch (tools/synth/synth_ref.cpp)
    if syn.synth_markov(C.c_void_p(ref.data_ptr()), C.c_int64(L), C.c_int64(S), C.c_void_p(cdf.data_ptr()), C.c_int(order), C.c_uint64(seed)) != 0:
        raise RuntimeError("synthetic reference kernel failed")
    rng = np.random.default_rng(seed)
    planted, n_fam = 0, 0
    while planted < 0.05 * L:
        flen = int(rng.integers(200, 3000))
        copies = int(min(10 ** rng.uniform(1, 4), max(10, 0.01 * L / flen)))
        fam = torch.randint(0, 4, (flen,), dtype=torch.uint8, device=dev, generator=g)
        c = fam[None, :].repeat(copies, 1)
        div = torch.rand(copies, 1, device=dev, generator=g) * 0.1
        m = torch.rand(copies, flen, device=dev, generator=g) < div
        c = torch.where(m, torch.randint(0, 4, (copies, flen), dtype=torch.uint8, device=dev, generator=g), c)
        rev = torch.rand(copies, device=dev, generator=g) < 0.5
        c = torch.where(rev[:, None], (3 - c).flip(1), c)
        pos = (torch.rand(copies, device=dev, generator=g, dtype=torch.float64) * (L - flen)).long()
        pos, _ = torch.sort(pos)                              # copies of one family must not overlap: an indexed store with duplicate
        keep = torch.ones(copies, dtype=torch.bool, device=dev)   # targets has no defined winner, and the reference must be the same every run
        keep[1:] = (pos[1:] - pos[:-1]) >= flen
        last = torch.cummax(torch.where(keep, pos, torch.zeros_like(pos)), 0)[0]   # start of the last kept copy at or before each one
        keep[1:] &= (pos[1:] - last[:-1]) >= flen
        pos, c = pos[keep], c[keep]
        ref[(pos[:, None] + torch.arange(flen, device=dev)[None, :]).reshape(-1)] = c.reshape(-1)
        planted += int(pos.numel()) * flen
        n_fam += 1
    return ref, lens, n_fam


def simulate_pairs(ref, lens, n_pairs, rl, seed, dev, ins_mean=400, ins_std=50, err=0.005, dup_frac=0.05, disc_frac=0.01, chim_frac=0.01, indel_frac=0.07):
    """Vectorised wgsim-like simulator on the GPU: returns uint8 codes [2*n_pairs, rl]."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), device=dev, dtype=torch.int64)
    lens_t = torch.tensor(lens, device=dev, dtype=torch.int64)
    ctg = torch.multinomial(lens_t.double() / lens_t.sum(), n_pairs, replacement=True, generator=g)
    d = (torch.randn(n_pairs, device=dev, generator=g) * ins_std + ins_mean).long().clamp(rl + 10, None)
    u = torch.rand(n_pairs, device=dev, generator=g)
    disc = (u >= dup_frac) & (u < dup_frac + disc_frac)
    d = torch.where(disc, torch.randint(5000, 50000, (n_pairs,), device=dev, generator=g), d)
    need = rl + 8
    d = torch.minimum(d, lens_t[ctg] - 1).clamp(need + 1, None)
    pos = (torch.rand(n_pairs, device=dev, generator=g, dtype=torch.float64) * (lens_t[ctg] - d).clamp(1, None).double()).long()
    dup = u < dup_frac
    src = (torch.rand(n_pairs, device=dev, generator=g, dtype=torch.float64) * torch.arange(n_pairs, device=dev).double()).long()
    ctg = torch.where(dup, ctg[src], ctg)
    pos = torch.where(dup, pos[src], pos)
    d = torch.where(dup, d[src], d)
    gpos = offs[ctg] + pos
    j = torch.arange(need, device=dev)
    f1 = ref[gpos[:, None] + j[None, :]]                                   # forward end
    f2 = 3 - ref[(gpos + d - 1)[:, None] - j[None, :]]                     # reverse-complemented far end
    chim = (~dup) & (~disc) & (torch.rand(n_pairs, device=dev, generator=g) < chim_frac)
    c2 = torch.multinomial(lens_t.double() / lens_t.sum(), n_pairs, replacement=True, generator=g)
    p2 = offs[c2] + (torch.rand(n_pairs, device=dev, generator=g, dtype=torch.float64) * (lens_t[c2] - need - 1).clamp(1, None).double()).long()
    other = ref[p2[:, None] + j[None, :]]
    bp = torch.randint(40, rl - 40, (n_pairs,), device=dev, generator=g)
    f1 = torch.where(chim[:, None] & (j[None, :] >= bp[:, None]), other, f1)
    flip = torch.rand(n_pairs, device=dev, generator=g) < 0.5
    r1 = torch.where(flip[:, None], f2, f1)
    r2 = torch.where(flip[:, None], f1, f2)
    reads = torch.stack([r1, r2], 1).reshape(2 * n_pairs, need)
    n = 2 * n_pairs
    # one indel of 1..5 bases in a fraction of the reads
    has = torch.rand(n, device=dev, generator=g) < indel_frac
    ip = torch.randint(5, rl - 5, (n,), device=dev, generator=g)
    il = torch.randint(1, 6, (n,), device=dev, generator=g)
    is_del = torch.rand(n, device=dev, generator=g) < 0.5
    jj = torch.arange(rl, device=dev)[None, :]
    src_del = jj + torch.where(jj >= ip[:, None], il[:, None], torch.zeros_like(jj))
    src_ins = jj - torch.where(jj >= (ip + il)[:, None], il[:, None], torch.zeros_like(jj))
    src_idx = torch.where((has & is_del)[:, None], src_del, torch.where((has & ~is_del)[:, None], src_ins, jj.expand(n, rl))).clamp(0, need - 1)
    out = torch.gather(reads, 1, src_idx)
    ins_mask = (has & ~is_del)[:, None] & (jj >= ip[:, None]) & (jj < (ip + il)[:, None])
    out = torch.where(ins_mask, torch.randint(0, 4, (n, rl), dtype=torch.uint8, device=dev, generator=g), out)
    sub = torch.rand(n, rl, device=dev, generator=g) < err
    out = torch.where(sub, (out + torch.randint(1, 4, (n, rl), dtype=torch.uint8, device=dev, generator=g)) % 4, out)
    nmask = torch.rand(n, rl, device=dev, generator=g) < 0.001
    out = torch.where(nmask, torch.full_like(out, 4), out)
    return out.contiguous()


def bwa_batches(n_pairs, rl, threads, chunk=10000000):
    """Upstream main_mem batching: reads are taken until >= chunk*threads bases with an even count."""
    per = -(-chunk * threads // (2 * rl))           # pairs per batch (ceil)
    pb = np.arange(n_pairs, dtype=np.int64) // per
    return pb.astype(np.int32), int(pb[-1]) + 1


def gpu_sample_sam(lib, idx, opt, hs, hoff, names, contig_names, lens):
    """The product path for the parity sample: ssg_mem_process_pairs -> ssg_sam_format -> ssg_sbl_markdup (all through the C ABI)."""
    import common
    from speedseq_amd import capi
    res = capi.mem_process_pairs(lib, idx, opt, hs, hoff, id0=0)
    text, _ = capi.sam_format(lib, idx, opt, res, names, hs, hoff, None, "")
    res.close()
    dup = capi.sbl_markdup(lib, common.sam_primary_ends(text, contig_names))
    return text, dup


def write_fastq(path, reads_np, rl, first_pair=0):
    """interleaved FASTQ of the uint8 codes [2n, rl]; fixed-width names p%09d, constant quality"""
    n2 = reads_np.shape[0]
    name_w = 10
    rec = np.empty((n2, 1 + name_w + 1 + rl + 3 + rl + 1), dtype=np.uint8)
    rec[:, 0] = ord("@")
    rec[:, 1] = ord("p")
    ids = (np.arange(n2, dtype=np.int64) // 2) + first_pair
    for k in range(9):
        rec[:, 2 + k] = ord("0") + (ids // 10 ** (8 - k)) % 10
    c = 1 + name_w
    rec[:, c] = 10
    rec[:, c + 1:c + 1 + rl] = np.frombuffer(b"ACGTN", dtype=np.uint8)[reads_np]
    rec[:, c + 1 + rl] = 10
    rec[:, c + 2 + rl] = ord("+")
    rec[:, c + 3 + rl] = 10
    rec[:, c + 4 + rl:c + 4 + 2 * rl] = ord("I")
    rec[:, c + 4 + 2 * rl] = 10
    rec.tofile(path)


def fastq_records_dev(reads, rl, first_pair=0):
    """write_fastq's records built on the device the reads are on: [2n, record bytes] uint8 (the host only copies and writes them -- the
    streamed soak feeds the pipeline at more than a million pairs a second from one thread of a 16-CPU box)"""
    import torch
    n2 = reads.shape[0]
    dev = reads.device
    name_w = 10
    rec = torch.empty((n2, 1 + name_w + 1 + rl + 3 + rl + 1), dtype=torch.uint8, device=dev)
    rec[:, 0] = ord("@")
    rec[:, 1] = ord("p")
    ids = torch.div(torch.arange(n2, dtype=torch.int64, device=dev), 2, rounding_mode="floor") + first_pair
    for k in range(9):
        rec[:, 2 + k] = (ord("0") + torch.div(ids, 10 ** (8 - k), rounding_mode="floor") % 10).to(torch.uint8)
    c = 1 + name_w
    rec[:, c] = 10
    lut = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=dev)
    rec[:, c + 1:c + 1 + rl] = lut.index_select(0, reads.reshape(-1).to(torch.int32)).reshape(n2, rl)   # int32 indices: a quarter of the int64 copy advanced indexing would make
    rec[:, c + 1 + rl] = 10
    rec[:, c + 2 + rl] = ord("+")
    rec[:, c + 3 + rl] = 10
    rec[:, c + 4 + rl:c + 4 + 2 * rl] = ord("I")
    rec[:, c + 4 + 2 * rl] = 10
    return rec


def e2e_leg(a, td, prefix, reads, reads2, rl, ns, orc_exe, b=None, reads_long=None, n_wide=100000):
    """The plugin path as the reference wires it (bin/speedseq:438-439): FASTQ file -> `bwa mem -t T -p` | `samblaster --excludeDups
    --addMateTags --maxSplitCount 2 --minNonOverlap 20 --splitterFile --discordantFile` -> three SAM streams on files, wall clock.
    The index is loaded from the files written by ssg_index_save; the rate excludes that one-off load (reported separately).
    A sample of the same FASTQ goes through the oracle's executables and the three streams must be byte-identical (modulo @PG)."""
    import subprocess
    import re
    bwa, sbl = (b("bwa"), b("samblaster")) if b else (os.path.join(ROOT, "bin", "bwa"), os.path.join(ROOT, "bin", "samblaster"))
    rn = reads.cpu().numpy()
    if reads2 is not None:
        rn = np.concatenate([rn, reads2.numpy()])      # the timed batch + a second one: several device calls, so the stages overlap
    fq = os.path.join(td, "reads.fq")
    write_fastq(fq, rn, rl)
    sfq = os.path.join(td, "sample.fq")
    write_fastq(sfq, rn[:2 * ns], rl)
    res = {"pairs": int(rn.shape[0] // 2), "bwa_threads": a.bwa_threads, "fastq": "uncompressed interleaved file in /dev/shm, 3 SAM streams written to /dev/shm"}

    def run(bwa_cmd, sbl_cmd, fastq, tag, threads):
        o, sp, di = (os.path.join(td, tag + x) for x in (".sam", ".spl.sam", ".disc.sam"))
        cmd = "%s mem -t %d -p %s %s 2> %s.bwa.err | %s --excludeDups --addMateTags --maxSplitCount 2 --minNonOverlap 20 --splitterFile %s --discordantFile %s > %s 2> %s.sbl.err" % (
            bwa_cmd, threads, prefix, fastq, os.path.join(td, tag), sbl_cmd, sp, di, o, os.path.join(td, tag))
        t = time.perf_counter()
        try:
            rc = subprocess.call(["bash", "-c", "set -o pipefail; " + cmd], timeout=300)
        except subprocess.TimeoutExpired:
            raise RuntimeError("pipeline %s: no result within 300 s" % tag)
        t = time.perf_counter() - t
        err = open(os.path.join(td, tag + ".bwa.err")).read()
        if rc != 0:
            raise RuntimeError("pipeline %s failed rc=%d: %s %s" % (tag, rc, err[-500:], open(os.path.join(td, tag + ".sbl.err")).read()[-500:]))
        return t, err, (o, sp, di)

    t, err, files = run(bwa, sbl, fq, "full", a.bwa_threads)
    m = re.search(r"wall: index load ([0-9.]+) s, reads -> SAM ([0-9.]+) s", err)
    mb = re.search(r"stage busy time: (.*)", err)
    if mb:
        res["bwa_stage_busy"] = mb.group(1)
    t_load, t_run = (float(m.group(1)), float(m.group(2))) if m else (None, None)
    res.update({"wall_s": round(t, 2), "index_load_s": t_load, "reads_to_sam_s": t_run,
                "pairs_per_s": res["pairs"] / (t - t_load) if t_load is not None else res["pairs"] / t,
                "pairs_per_s_incl_index_load": res["pairs"] / t,
                "sam_bytes": os.path.getsize(files[0]), "splitter_bytes": os.path.getsize(files[1]), "discordant_bytes": os.path.getsize(files[2])})
    if os.environ.get("SSG_E2E_STAGE_LOG"):              # diagnostics: the same run with the library's per-stage wall times (adds a sync per stage)
        os.environ["SSG_DEBUG"] = "1"
        os.environ["SSG_BWA_PROF"] = "1"                 # ... and its per-kernel device time over the whole run
        try:
            _, errd, _ = run(bwa, sbl, fq, "dbg", a.bwa_threads)
            open(os.environ["SSG_E2E_STAGE_LOG"], "w").write(errd)
        finally:
            del os.environ["SSG_DEBUG"]
            del os.environ["SSG_BWA_PROF"]
    # gz input (the reference pipeline reads .fq.gz): one inflate stream bounds the ingest
    n_gz = min(res["pairs"], 2000000)                     # fixed-width records: a byte prefix is a whole number of pairs
    rec_bytes = os.path.getsize(fq) // (2 * res["pairs"])
    if subprocess.call(["bash", "-c", "head -c %d %s | gzip -1 -c > %s.gz" % (2 * n_gz * rec_bytes, fq, fq)]) == 0:
        tg, errg, _ = run(bwa, sbl, fq + ".gz", "fullgz", a.bwa_threads)
        mg = re.search(r"wall: index load ([0-9.]+) s", errg)
        res["pairs_per_s_gz_input"] = n_gz / (tg - float(mg.group(1))) if mg else n_gz / tg
        res["gz_input_pairs"] = n_gz
    # parity of the executables on the sample
    _, _, gf = run(bwa, sbl, sfq, "s_gpu", a.bwa_threads)
    t_orc, err_orc, of = run(orc_exe, orc_exe + " samblaster", sfq, "s_orc", min(os.cpu_count() or 1, 64))   # one upstream batch either way: -t only sets the worker count
    res["oracle_cli"] = {"what": "the same command line on the oracle's executables (scalar C port), FASTQ file -> three SAM streams, index load included",
                         "pairs": ns, "threads": min(os.cpu_count() or 1, 64), "wall_s": round(t_orc, 2), "pairs_per_s": ns / t_orc}

    def nopg(p):
        return [l for l in open(p).read().split("\n") if not l.startswith("@PG")]
    same = [nopg(x) == nopg(y) for x, y in zip(gf, of)]
    res["sample_pairs"] = ns
    res["sample_streams_identical"] = bool(all(same))
    res["sample_streams"] = {"sam": same[0], "splitters": same[1], "discordants": same[2]}
    for f in os.listdir(td):                              # the multi-GB SAM streams of the timed runs: free /dev/shm for what follows
        if f.startswith(("full.", "fullgz.", "dbg.")) or f.endswith(".fq.gz"):
            os.remove(os.path.join(td, f))
    # ---- parity where it used to be thin: on THIS index, product executables against the oracle's, byte for byte:
    # other option letters, single-end input, 2x250 -- n_wide pairs (reads) each
    nw = min(n_wide, int(rn.shape[0] // 2))
    wide = {}
    if nw > 0:
        wfq = os.path.join(td, "wide.fq")
        write_fastq(wfq, rn[:2 * nw], rl)
        legs = [("pe_default", "-p", wfq, True), ("pe_M_Y", "-p -M -Y", wfq, True), ("pe_scores_A2_B5_O7,9_E2,1", "-p -A 2 -B 5 -O 7,9 -E 2,1", wfq, True),
                ("pe_k25_T40", "-p -k 25 -T 40", wfq, True), ("single_end", "", wfq, False)]
        if reads_long is not None:
            nl = min(nw, int(reads_long.shape[0] // 2))
            lfq = os.path.join(td, "wide_long.fq")
            write_fastq(lfq, reads_long[:2 * nl].cpu().numpy(), int(reads_long.shape[1]))
            legs.append(("pe_2x%d" % int(reads_long.shape[1]), "-p", lfq, True))

        def run_opts(bwa_cmd, sbl_cmd, fastq, tag, threads, opts, paired):
            o, sp, di = (os.path.join(td, tag + x) for x in (".sam", ".spl.sam", ".disc.sam"))
            if paired:
                cmd = "%s mem -t %d %s %s %s 2> %s.bwa.err | %s --excludeDups --addMateTags --maxSplitCount 2 --minNonOverlap 20 --splitterFile %s --discordantFile %s > %s 2> %s.sbl.err" % (
                    bwa_cmd, threads, opts, prefix, fastq, os.path.join(td, tag), sbl_cmd, sp, di, o, os.path.join(td, tag))
            else:   # single-end reads: `bwa mem` alone (samblaster is a paired-end tool in the reference's pipeline)
                cmd = "%s mem -t %d %s %s %s 2> %s.bwa.err > %s" % (bwa_cmd, threads, opts, prefix, fastq, os.path.join(td, tag), o)
            rc = subprocess.call(["bash", "-c", "set -o pipefail; " + cmd], timeout=300)
            if rc != 0:
                raise RuntimeError("pipeline %s failed rc=%d: %s" % (tag, rc, open(os.path.join(td, tag + ".bwa.err")).read()[-400:]))
            return (o, sp, di) if paired else (o,)
        for name, opts, fastq, paired in legs:
            try:
                gfw = run_opts(bwa, sbl, fastq, "w_gpu", a.bwa_threads, opts, paired)
                ofw = run_opts(orc_exe, orc_exe + " samblaster", fastq, "w_orc", min(os.cpu_count() or 1, 64), opts, paired)
                gl, ol = [nopg(x) for x in gfw], [nopg(y) for y in ofw]
                wide[name] = {"options": opts.strip(), "pairs" if paired else "reads": (nw if fastq == wfq else nl) * (1 if paired else 2), "sam_lines": len(gl[0]),
                              "identical": bool(gl == ol), "lines_differing": sum(1 for x, y in zip(gl[0], ol[0]) if x != y) + abs(len(gl[0]) - len(ol[0]))}
                for x in gfw + ofw:
                    os.remove(x)
            except Exception as e:
                wide[name] = {"options": opts.strip(), "error": repr(e)[:300]}
        res["parity_wide"] = {"what": "bin/bwa mem OPTIONS | bin/samblaster (the reference's switches) against oracle/orc_bwa mem OPTIONS | orc_bwa samblaster on the same FASTQ and the same "
                                      "index files: the three SAM streams byte for byte (@PG aside); single-end: `bwa mem` alone", "legs": wide,
                              "all_identical": bool(all(v.get("identical") for v in wide.values()))}
    return res


def script_leg(td, tag, ref_prefix, fq, n_pairs, threads, bwa, samblaster, sambamba, sort_mem_gb=40, config_extra="", env_extra=None, limit_s=300, ranks=0):
    """SURVEY.md 8d's own definition of the metric: the reference's `speedseq align` (the script itself, unmodified -- the fixture copy
    tests/golden/speedseq_ref_script.sh, or /root/reference/bin/speedseq where that exists) on the executables speedseq.config names,
    wall clock from FASTQ open to the three coordinate-sorted, indexed BAMs closed.  The index files are already next to `ref_prefix`.
    config_extra: further lines of speedseq.config (the script `source`s it: `export SSG_FUSED=1` switches the product's stages to
    the binary hand-off of speedseq_amd/host/fused.h)."""
    import shutil
    import subprocess
    script = "/root/reference/bin/speedseq" if os.path.exists("/root/reference/bin/speedseq") else os.path.join(ROOT, "tests", "golden", "speedseq_ref_script.sh")
    awk = shutil.which("gawk") or shutil.which("mawk") or shutil.which("awk")
    if not os.path.exists(script) or awk is None:
        return {"skipped": "reference script fixture or awk not available"}
    d = os.path.join(td, "script_" + tag)
    os.makedirs(d)
    # the wrappers must be executable: /dev/shm (where the data lives) is mounted noexec on the GPU box, so they go next to the repository's own executables
    bindir = os.path.join(ROOT, "gpurun_out", ".wrap_%d_%s" % (os.getpid(), tag))
    shutil.rmtree(bindir, ignore_errors=True)
    os.makedirs(bindir)
    if not shutil.which("gawk"):
        os.symlink(awk, os.path.join(bindir, "gawk"))           # the script hard-codes `gawk`
    for name, cmd in (("bwa", bwa), ("samblaster", samblaster)):  # wrappers: a config entry must be one file, the oracle's samblaster is `orc_bwa samblaster`
        with open(os.path.join(bindir, name), "w") as f:
            f.write("#!/bin/sh\nexec %s \"$@\"\n" % cmd)
        os.chmod(os.path.join(bindir, name), 0o755)
    if not os.path.exists(ref_prefix):
        open(ref_prefix, "w").write(">placeholder: the five index files next to this name are what bwa mem reads\n")
    cfg = os.path.join(d, "speedseq.config")
    open(cfg, "w").write("BWA=%s/bwa\nSAMBLASTER=%s/samblaster\nSAMBAMBA=%s\nPARALLEL=%s/bin/parallel\n%s" % (bindir, bindir, sambamba, ROOT, config_extra))
    out = os.path.join(d, "out")
    env = dict(os.environ, PATH="%s:%s" % (bindir, os.environ["PATH"]))
    env.update(env_extra or {})
    import signal
    t = time.perf_counter(); t_epoch = time.time()
    launcher = [os.path.join(ROOT, "bin", "speedseq-ranks"), "-n", str(ranks), "--script", script, "--"] if ranks > 1 else ["bash", script]   # rank mode: N pipelines side by side (speedseq_amd/host/ranks.h)
    p = subprocess.Popen(launcher + ["align", "-K", cfg, "-o", out, "-M", str(sort_mem_gb), "-t", str(threads), "-p",
                          "-R", "@RG\\tID:bench\\tSM:bench\\tLB:lib1", ref_prefix, fq], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        so, se = p.communicate(timeout=limit_s)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)                       # the whole pipeline (the script's children share the session)
        so, se = p.communicate()
        shutil.rmtree(bindir, ignore_errors=True)
        for f in os.listdir("/dev/shm") if os.path.isdir("/dev/shm") else []:   # frame payloads of the killed pipeline (fused.h segments): their writers are gone
            m = f.split(".")
            if f.startswith("ssgfuse.") and len(m) == 3 and m[1].isdigit() and not os.path.exists("/proc/" + m[1]):
                try:
                    os.remove(os.path.join("/dev/shm", f))
                except OSError:
                    pass
        return {"error": "no result within %d s; stderr tail: %s" % (limit_s, se[-600:])}
    t = time.perf_counter() - t
    shutil.rmtree(bindir, ignore_errors=True)
    r = type("R", (), {"returncode": p.returncode, "stdout": so, "stderr": se})
    if r.returncode != 0:
        keep = os.path.join(ROOT, "gpurun_out", "script_%s_failed.stderr" % tag)   # the whole of it: the tail is the script's own echo of its commands
        try:
            os.makedirs(os.path.dirname(keep), exist_ok=True)
            open(keep, "w").write(r.stderr)
        except OSError:
            pass
        msgs = [l for l in r.stderr.split("\n") if l.startswith(("[bwa]", "[samblaster]", "[sambamba]", "[ssgpu]", "[ranks]")) and not l.startswith(("[bwa] wall", "[bwa] stage busy", "[sambamba] sort:"))]
        return {"error": "rc %d after %.0f s; %s ... %s" % (r.returncode, t, " | ".join(msgs[-6:])[:1200], r.stderr[-300:]), "wall_s": round(t, 2)}
    sizes = {x: os.path.getsize(out + x) for x in (".bam", ".splitters.bam", ".discordants.bam")}
    ok = all(os.path.exists(out + x + ".bai") for x in sizes)
    stages = [l for l in r.stderr.split("\n") if l.startswith(("[bwa] wall", "[bwa] stage busy", "[sambamba] sort:", "[samblaster] pairs", "[samblaster] main thread", "[samblaster] first stage", "[ssgpu] index load", "[ssgpu] device arena", "[bwa] kernel", "[bwa] device"))]
    # SSG_STAMP=1 (config_extra): when each stage started and ended, in seconds after the script was launched
    stamps = []
    for l in r.stderr.split("\n"):
        if l.startswith("[stamp] "):
            w = l.split()
            stamps.append((round(float(w[3]) - t_epoch, 3), w[1], w[2]))
    res = {"pairs": n_pairs, "threads": threads, "wall_s": round(t, 2), "pairs_per_s": n_pairs / t, "bam_bytes": sizes, "bai_written": ok, "out": out, "stage_log": stages}
    if stamps:
        res["timeline_s"] = ["%.3f %s %s" % x for x in sorted(stamps)]
    return res


def bam_view(samtools, bam):
    """decoded records + header (modulo @PG) of a BAM through the reference's samtools (oracle/_ref, built from /root/reference/src/samtools-1.3.1)"""
    import subprocess
    txt = subprocess.check_output([samtools, "view", "-h", bam], text=True)
    return [l for l in txt.split("\n") if not l.startswith("@PG")]


def literal_legs(a, td, prefix, rl, ns, b, orc_exe):
    """The metric as SURVEY.md 8d words it -- wall clock, FASTQ open -> three coordinate-sorted BAMs + BAI closed -- through the reference's own
    script (unmodified) on the product executables: text hand-off (the parity path) on a prefix of the file, fused hand-off (SSG_FUSED=1 in
    speedseq.config) on --script-pairs pairs; the fused run's BAMs of the sample must decode to the records of the oracle's run; the CPU
    baseline is the same script on the oracle's executables + the reference's samtools."""
    fq = os.path.join(td, "reads.fq")
    rec_bytes = 2 * (1 + 10 + 1 + rl + 3 + rl + 1)                      # bytes per pair of write_fastq's fixed-width records
    n_all = os.path.getsize(fq) // rec_bytes
    samtools = os.path.join(ROOT, "oracle", "_ref", "samtools")
    shim = os.path.join(ROOT, "tools", "sambamba_samtools_shim.sh")
    host_cfg = "export SSG_SORT_THREADS=%d\nexport SSG_FMT_THREADS=%d\nexport SSG_SORT_LOG=1\nexport SSG_SBL_LOG=1\nexport SSG_LOAD_LOG=1\n" % (min(os.cpu_count() or 8, 256), min(os.cpu_count() or 8, 48))   # the script's -t sizes upstream's batches; the host pools are sized for the box
    host_cfg += os.environ.get("SSG_BENCH_CONFIG_EXTRA", "").replace(";", "\n") + "\n"   # A/B runs: further `export X=Y` lines for speedseq.config
    fused_cfg = "export SSG_FUSED=1\n" + host_cfg
    res = {"metric": "paired reads aligned+dup-marked/sec, FASTQ file -> out.bam + out.splitters.bam + out.discordants.bam (+ .bai), `speedseq align` wall clock incl. index load"}

    def head(n, name):
        p = os.path.join(td, name)
        with open(fq, "rb") as f, open(p, "wb") as g:
            left = n * rec_bytes
            while left:
                blk = f.read(min(left, 64 << 20)); g.write(blk); left -= len(blk)
        return p
    n_fused = min(a.script_pairs, n_all)
    fq_f = fq if n_fused == n_all else head(n_fused, "fused.fq")
    r = script_leg(td, "fused", prefix, fq_f, n_fused, a.script_threads, b("bwa"), b("samblaster"), b("sambamba"), config_extra=fused_cfg, limit_s=120)
    if "pairs_per_s" in r:   # the GPU boxes are shared: the same command took 4.4 .. 7.1 s on different boxes and within one session (profiles/r06y_literal_repeats.json).  Another tenant's process on the same GPU (rocm-smi --showpids lists it) makes about one run in three a second slower.  Run three times, report the fastest, keep all
        walls = [r.get("wall_s")]
        for k in (2, 3):
            r2 = script_leg(td, "fused%d" % k, prefix, fq_f, n_fused, a.script_threads, b("bwa"), b("samblaster"), b("sambamba"), config_extra=fused_cfg, limit_s=120)
            walls.append(r2.get("wall_s"))
            if "pairs_per_s" in r2 and r2["pairs_per_s"] > r["pairs_per_s"]:
                r = r2
        r["repeats_wall_s"] = walls
    if "pairs_per_s" not in r:   # frame payloads as mapped segments are the newest part of the hand-off: once more with every payload on the pipes before giving the number to the text path
        first_error = r.get("error")
        r = script_leg(td, "fused", prefix, fq_f, n_fused, a.script_threads, b("bwa"), b("samblaster"), b("sambamba"), config_extra=fused_cfg + "export SSG_FUSED_SHM=0\n", limit_s=120)
        r["segments_run_failed"] = first_error
    r.pop("out", None)
    r["what"] = "`speedseq align -t %d -p` (the reference's script, unmodified) on bin/bwa, bin/samblaster, bin/sambamba with `export SSG_FUSED=1` in speedseq.config (binary hand-off between the stages, speedseq_amd/host/fused.h)" % a.script_threads
    res["fused"] = r
    log('script, fused hand-off: %s pairs in %s s' % (r.get('pairs'), r.get('wall_s')))
    res["value"] = r.get("pairs_per_s")
    res["value_from"] = "fused"
    res["unit"] = "pairs/s"
    fused_ok = "pairs_per_s" in r
    n_text = min(2000000, n_all) if fused_ok else n_fused            # the text hand-off carries the literal number when the fused run gave none
    r = script_leg(td, "text", prefix, fq_f if n_text == n_fused else head(n_text, "text.fq"), n_text, a.script_threads, b("bwa"), b("samblaster"), b("sambamba"),
                   config_extra=host_cfg)
    r.pop("out", None)
    r["what"] = "the same without SSG_FUSED: SAM text on every pipe (the parity path)"
    res["text"] = r
    if not fused_ok:
        res["value"], res["value_from"] = r.get("pairs_per_s"), "text"
    log('script, text hand-off: %s pairs in %s s' % (r.get('pairs'), r.get('wall_s')))
    if os.path.exists(os.path.join(td, "text.fq")):
        os.remove(os.path.join(td, "text.fq"))
    # parity of the fused path on the sample: product (fused) vs the oracle's executables behind the same script
    sfq = os.path.join(td, "sample.fq")
    if os.path.exists(samtools) and os.path.exists(sfq):
        rp = script_leg(td, "s_fused", prefix, sfq, ns, a.script_threads, b("bwa"), b("samblaster"), b("sambamba"), config_extra=fused_cfg if fused_ok else host_cfg, limit_s=120)
        ro = script_leg(td, "s_orc", prefix, sfq, ns, a.script_threads, orc_exe, orc_exe + " samblaster", shim, sort_mem_gb=8)
        if "out" in rp and "out" in ro:
            same = {x: bam_view(samtools, rp["out"] + x) == bam_view(samtools, ro["out"] + x) for x in (".bam", ".splitters.bam", ".discordants.bam")}
            res["sample_bams_equal_oracle"] = bool(all(same.values()))
            res["sample_bams"] = dict(same, pairs=ns, what="samtools view -h of the three BAMs (header modulo @PG): product run (%s hand-off) vs the oracle's executables + the reference's samtools behind the same script" % ("fused" if fused_ok else "text"))
        else:
            res["sample_bams"] = {"error": (rp.get("error") or "") + (ro.get("error") or "")}
    # CPU baseline, BASELINE.md section 3: `speedseq align -t <hardware threads>` on the oracle's executables, on enough pairs that the fixed costs
    # (the oracle's 5.4 GB index load) no longer dominate: one short run first (page cache, allocator), then the measured run
    if a.cpu_script_pairs > 0 and os.path.exists(samtools) and time.time() - _T0 < 420:
        nc = min(a.cpu_script_pairs, n_all)
        cores = min(os.cpu_count() or 1, 128)
        wfq = head(min(nc, 50000), "cpuw.fq")
        rw = script_leg(td, "cpuw", prefix, wfq, min(nc, 50000), cores, orc_exe, orc_exe + " samblaster", shim, sort_mem_gb=8)
        log('script on the oracle executables, warm-up on %d pairs: %s' % (min(nc, 50000), rw.get('wall_s', rw.get('error'))))
        cfq = head(nc, "cpu.fq")
        rc = script_leg(td, "cpu0", prefix, cfq, nc, cores, orc_exe, orc_exe + " samblaster", shim, sort_mem_gb=8, limit_s=400)
        log('script on the oracle executables, %d pairs: %s' % (nc, rc.get('wall_s', rc.get('error'))))
        if "wall_s" in rc:
            fixed = rw.get("wall_s")
            res["cpu_script"] = {"value": nc / rc["wall_s"], "unit": "pairs/s", "cores": cores, "threads_given": cores, "hardware_threads": os.cpu_count(), "host_cpu_quota": host_cpu_quota(), "kind": "port", "pairs": nc, "wall_s": rc["wall_s"],
                                 "warmup_run": {"pairs": min(nc, 50000), "wall_s": fixed},
                                 "marginal_pairs_per_s": (nc - min(nc, 50000)) / (rc["wall_s"] - fixed) if fixed and rc["wall_s"] > fixed and nc > 50000 else None,
                                 "what": "`speedseq align -t %d -p` (the reference's script, unmodified) with oracle/orc_bwa as bwa and samblaster and the reference's samtools 1.3.1 behind sambamba's command line, "
                                         "FASTQ -> three sorted BAMs + BAI, index files pre-built and in the page cache (a 50 k-pair run first), index load included in the wall time; "
                                         "marginal_pairs_per_s = the rate between the two runs, i.e. without the fixed costs" % cores}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1000000, help="pairs per GPU per step (BASELINE.json configs[1]: 1M)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --pairs is the TOTAL over all ranks (each aligns pairs / N of one input); default is weak scaling, --pairs per GPU as the bench contract runs it")
    ap.add_argument("--ref-mbp", type=float, default=3100.0, help="synthetic GRCh37-shaped reference size (GRCh37 = 3100)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--ins-mean", type=int, default=0, help="fragment length mean (0: 400, or 800 for reads of 250 bases and more -- BASELINE.json configs[4]: wgsim -d 800 -s 150)")
    ap.add_argument("--ins-std", type=int, default=0)
    ap.add_argument("--bwa-threads", type=int, default=16, help="the -t whose batch boundaries (insert-size model scope) are reproduced")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="pairs of the timed batch aligned by the CPU oracle: parity gate on the timed call + cpu_baseline (-1 = the whole batch, 0 = skip)")
    ap.add_argument("--script-pairs", type=int, default=8000000, help="pairs in the FASTQ of the literal-metric leg: the reference's `speedseq align` script on the product executables, FASTQ -> three sorted BAMs + BAI")
    ap.add_argument("--script-threads", type=int, default=16, help="-t of the script legs on the product executables")
    ap.add_argument("--cpu-script-pairs", type=int, default=1000000, help="pairs of the CPU baseline through the script (`speedseq align -t <cores>` on the oracle's executables; 0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--profile", dest="no_profile", action="store_false", help="(with --emu-selftest, which switches the per-kernel timing off) keep it on")
    ap.add_argument("--no-e2e", dest="e2e", action="store_false", help="skip the plugin-path leg (bin/bwa mem | bin/samblaster on FASTQ files)")
    ap.add_argument("--e2e-pairs", type=int, default=4000000, help="pairs in the FASTQ file of the plugin-path leg (the timed batch + freshly simulated ones): enough device calls for the pipeline's steady state to show")
    ap.add_argument("--config5-pairs", type=int, default=200000, help="pairs of the 2x250 leg (BASELINE.json configs[4]: long fragments, wider bands), 0 = skip")
    ap.add_argument("--wide-pairs", type=int, default=100000, help="pairs (reads) of each leg of the wide parity check of the plugin-path leg: option letters, single-end, 2x250 against the oracle's executables on this index; 0 = skip")
    ap.add_argument("--no-dist-rehearsal", dest="dist_rehearsal", action="store_false", help="skip the one-rank rehearsal of the N > 1 code path (a second process)")
    ap.add_argument("--emu-selftest", action="store_true", help="TEST INFRASTRUCTURE, never a measurement: walk this script's whole flow on the CPU with the host-emulation build "
                    "(tests/emu) and the bundled chr20 slice as the reference, at toy sizes -- catches a broken leg before it costs GPU minutes")
    ap.add_argument("--partial", default=os.path.join(ROOT, "gpurun_out", "bench_partial.json"), help="the line so far is also written here after every leg (a run that is cut short leaves its numbers)")
    if "--emu-selftest" in sys.argv:                         # toy sizes unless given: the emulation runs the kernels lane by lane on the CPU
        ap.set_defaults(pairs=600, steps=1, warmup=0, bwa_threads=1, e2e_pairs=1200, script_pairs=1200, script_threads=2, cpu_script_pairs=200, config5_pairs=300, wide_pairs=300,
                        no_profile=True, partial="/tmp/bench_emu_partial.json")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = world == 1 and os.environ.get("SSG_BENCH_FORCE_DIST") == "1"   # single-rank rehearsal of the N>1 code path
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    multi = world > 1 or force_dist
    saved_stdout = None
    if multi:
        # RCCL prints a version banner (and warnings) straight to stdout; this program's stdout is ONE JSON line, so everything
        # else is sent to stderr until that line is written
        sys.stdout.flush(); saved_stdout = os.dup(1); os.dup2(2, 1)
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/rccl.%h.%p.log")   # RCCL's banner / warnings off stdout: this program prints ONE JSON line
        import torch.distributed as dist
        from speedseq_amd import dist as ssdist
        if a.emu_selftest:
            dist.init_process_group("gloo")                   # the flow check of the N > 1 step on the host emulation
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if a.strong and world > 1:
        a.pairs = max(1, a.pairs // world)          # this rank's share; the ranks' reads are simulated with different seeds (12 + rank): one input of world * a.pairs pairs
    emu = a.emu_selftest
    if not emu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path")
    if emu:
        class _NoCuda:                                       # the flow below calls torch.cuda.* between legs
            def __getattr__(self, _):
                return lambda *x, **k: None
        torch.cuda = _NoCuda()
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)

    from speedseq_amd import capi
    lib = capi.Lib(os.path.join(ROOT, "tests", "emu", "libssgpu_emu.so") if emu else None)
    lib._chk(lib.l.ssg_set_device(C.c_int(local)))
    opt = lib.opt_init()

    t0 = time.time()
    if emu:
        import simreads
        codes = np.concatenate([c for _, c in simreads.read_fasta(os.path.join(ROOT, "tests", "golden", "chr20_slice.fa"))]).astype(np.uint8)
        codes[codes > 3] = 0
        ref, lens, n_fam = torch.from_numpy(codes), [int(codes.size)], 0
    else:
        ref, lens, n_fam = synth_reference(int(a.ref_mbp * 1e6), 20150810, dev)
    torch.cuda.synchronize()
    t_ref = time.time() - t0
    ctg_off = np.concatenate([[0], np.cumsum(lens)])[:-1]
    t0 = time.time()
    torch.cuda.empty_cache()
    ctg_names = GRCH37_NAMES[:len(lens)]
    idx = lib.index_build_dev(ref.data_ptr(), int(ref.numel()), ctg_off, lens, ctg_names)   # `bwa index` on the device (k_index.h)
    t_index = time.time() - t0
    log('reference (%.1f s) and index (%.1f s) on the device' % (t_ref, t_index))

    rl = a.read_len
    if a.cpu_sample < 0:
        a.cpu_sample = a.pairs
    ins = dict(ins_mean=a.ins_mean or (800 if rl >= 250 else 400), ins_std=a.ins_std or (150 if rl >= 250 else 50))
    reads = simulate_pairs(ref, lens, a.pairs, rl, 12 + rank, dev, **ins)
    n_more = max(0, max(a.e2e_pairs, a.script_pairs) - a.pairs)
    reads_e2e = simulate_pairs(ref, lens, n_more, rl, 1012, dev, **ins).cpu() if (a.e2e and world == 1 and a.cpu_sample > 0 and n_more > 0) else None   # further pairs for the plugin-path leg
    # N > 1: rank 0 also times the reference's script on `bin/bwa mem` over all N devices (the product's own multi-GPU path) once the step is measured
    n_ml = min(a.script_pairs, 4000000 * world) if (world > 1 and a.e2e and rank == 0) else 0
    reads_ml = simulate_pairs(ref, lens, n_ml, rl, 2012, dev, **ins).cpu() if n_ml > 0 else None
    reads5 = simulate_pairs(ref, lens, a.config5_pairs, 250, 512, dev, ins_mean=800, ins_std=150) if (a.config5_pairs > 0 and rl == 150 and world == 1 and a.cpu_sample > 0) else None
    d_seq = reads.reshape(-1)
    d_off = (torch.arange(2 * a.pairs + 1, device=dev, dtype=torch.int64) * rl).contiguous()
    pb, n_batches = bwa_batches(a.pairs, rl, a.bwa_threads)
    d_pb = torch.from_numpy(pb).to(dev)
    del ref
    torch.cuda.empty_cache()
    torch.cuda.synchronize()

    d_sig = torch.empty((a.pairs, 3), dtype=torch.int64, device=dev) if multi else None
    ordinal = (torch.arange(a.pairs, device=dev, dtype=torch.int64) + rank * a.pairs) if multi else None
    n_dup_global, merge_bytes = [0], [0]
    rec_bytes = capi.dev_record_bytes(lib)

    def step(want_dup=False):
        if not multi:
            return capi.hotpath_dev_ex(lib, idx, opt, a.pairs, rl, d_seq.data_ptr(), d_off.data_ptr(), d_pb.data_ptr(), n_batches, 0)
        # N > 1: every rank aligns its shard; duplicates are decided over the whole input (first-seen-wins on the global pair
        # ordinal): signatures routed to an owner rank by hash with one all-to-all, verdicts routed back (speedseq_amd/dist.py);
        # the side-stream classification (discordant / splitter, --excludeDups) then runs on the records still resident in HBM
        summary, h = capi.hotpath_dev_ex(lib, idx, opt, a.pairs, rl, d_seq.data_ptr(), d_off.data_ptr(), d_pb.data_ptr(), n_batches, 0,
                                         local_dedup=False, d_sig=d_sig.data_ptr(), keep=True)
        valid = d_sig[:, 0] != -1
        dup = ssdist.global_markdup(d_sig, valid, ordinal, device=dev, lib=lib).to(torch.uint8).contiguous()
        c = capi.dev_records_classify(lib, h, dup.data_ptr())
        # ... and the coordinate-sorted merge (coupling 3): keys + fixed-size records of every SAM line travel to the rank that owns
        # their key range (sample sort: one all-to-all), where they are sorted by (key, global ordinal)
        nl = capi.dev_records_n_lines(lib, h)
        d_keys = torch.empty(nl, dtype=torch.int64, device=dev)
        d_recs = torch.empty((nl, rec_bytes), dtype=torch.uint8, device=dev)
        capi.dev_records_export(lib, h, d_keys.data_ptr(), d_recs.data_ptr())
        capi.dev_records_free(lib, h)
        line_ord = torch.arange(nl, device=dev, dtype=torch.int64) + (rank << 40)       # global ordinal: rank-major input order
        _, _, _, sent = ssdist.coordinate_range_exchange(d_keys, line_ord, d_recs, device=dev)
        merge_bytes[0] = sent
        summary[1], summary[8], summary[9], summary[10] = c[0], c[1], c[2], c[3]
        n_dup_global[0] = int(c[0])
        return summary, dup

    for _ in range(a.warmup):
        step()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    prof = not a.no_profile
    if prof:
        lib.l.ssg_prof_reset()
        lib.l.ssg_prof_enable(C.c_int(1))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        summary, _ = step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    kern = capi.prof_get(lib) if prof else {}
    log('timed steps done: %.1f ms/step' % (1e3 * dt / a.steps))
    if prof:
        lib.l.ssg_prof_enable(C.c_int(0))

    workload = ("%d synthetic 2x%d bp PE pairs per GPU vs a seeded synthetic GRCh37-shaped reference of %.0f Mbp (25 contigs, order-6 Markov "
                "model of the bundled chr20 slice + %d planted repeat families of 10..10^4 copies at 0-10%% divergence = 5%% of the bases; "
                "GRCh37 itself is not available offline), bwa-mem -t %d batch boundaries" % (a.pairs, rl, sum(lens) / 1e6, n_fam, a.bwa_threads))
    out = {
        "metric": "paired reads aligned+dup-marked/sec", "value": world * a.pairs * a.steps / dt, "unit": "pairs/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
        "scaling": "strong" if a.strong else "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": workload,
                   "pairs_per_gpu": a.pairs, "read_len": rl, "fragment_mean_std": [ins["ins_mean"], ins["ins_std"]], "ref_bp": int(sum(lens)), "index_build_s": round(t_index, 2), "ref_synth_s": round(t_ref, 2),
                   "records": int(summary[0]), "dup_pairs": n_dup_global[0] if multi else int(summary[1]), "dup_pairs_local_view": int(summary[1]), "seeds": int(summary[2]), "rescues": int(summary[5]),
                   "bwt_extends": int(summary[6]), "chains": int(summary[7]),
                   "sam_lines": int(summary[10]), "discordant_stream_lines": int(summary[8]), "splitter_stream_lines": int(summary[9]),
                   "dedup_scope": ("global over all ranks (all-to-all signature exchange; owner side: %s)" % ssdist.OWNER_PATH[0]) if multi else "single GPU = whole input",
                   "sorted_merge": ("coordinate range exchange of %d-byte records by samtools' key inside the step; %.1f MB sent by rank 0 per step" % (rec_bytes, merge_bytes[0] / 1e6)) if multi else "single GPU: bin/sambamba sort (device radix sort of the keys)"},
    }
    def save_partial():
        try:
            os.makedirs(os.path.dirname(a.partial), exist_ok=True)
            with open(a.partial, "w") as f:
                json.dump(out, f)
        except Exception:
            pass
    if rank == 0:
        save_partial()
        # ---- roofline of the dominant kernel (live HIP-event timing on the launch stream) ----
        if kern:
            name, (ms, cnt) = max(kern.items(), key=lambda kv: kv[1][0])
            per_launch_ms = ms / cnt
            n_ext, seeds, recs, nreads = float(summary[6]), float(summary[2]), float(summary[0]), 2.0 * a.pairs
            # ALGORITHMIC bytes per launch of each hot kernel (DESIGN.md section 4 states the per-unit figures):
            alg_bytes = {
                # every bwt_extend = 2 rank queries = 2 x 64-byte Occ blocks (SURVEY 8d B_fm) + the read itself
                "ssg_k_smem_quad": 128.0 * n_ext + nreads * rl,
                # per seed: ~1.5 LF steps (SA sampled every 4 rows in HBM) x one 64-byte block + one 8-byte SA sample + the 28-byte seed written
                "ssg_k_sal": seeds * (1.5 * 64 + 8 + 28),
                # per seed: 28 bytes read (seed + contig id); per read <= one 56-byte chain + 4-byte id per seed written
                "ssg_k_chain": seeds * (28 + 56 + 4),
                # per read: the read, one 2-bit reference window per chain (~l+400 bases), <= one 88-byte region per seed
                "ssg_k_chain2aln": nreads * (rl + (rl + 400) / 4.0) + seeds * 28 + recs * 88,
                # per rescue: mate read + 2-bit window (~insert range + l) + one 88-byte region
                "ssg_k_matesw": float(summary[5]) * (rl + (rl + 500) / 4.0 + 88),
                "ssg_k_pair_final": recs * 88 * 2 + recs * 32,
                "ssg_k_reg2aln": recs * (88 + rl + (rl + 50) / 4.0 + 632),
                # per chain: both read sides + their 2-bit windows + 40-byte job + 2 x 32-byte results
                "ssg_k_ext_lane<136>": float(summary[7]) * (rl + (rl + 140) / 4.0 + 104) if len(summary) > 7 and summary[7] else 0.0,
            }
            # the seeding stage (rows a1-a2) is the HBM-bound part of the step -- random 64-byte rank blocks -- and the one the roofline line is about:
            # since round 4 it is two launches, the lane-per-read kernel and the wave-per-read kernel for the reads it gives up (k_smem2.h); the
            # algorithmic bytes are upstream's own count of bwt_extend calls x 2 rank blocks, whoever ran them.  The largest SINGLE kernel by
            # time is named in `largest_kernel`: when that is the mate-rescue Smith-Waterman, an on-chip integer DP, neither an HBM nor an MFMA
            # roofline applies to it and its figure of merit is the VALU issue fraction in `sw`.
            if "ssg_k_ext_lane_dyn<4>" in kern:   # the extension kernel with per-class LDS (all classes above 72 columns in one name)
                alg_bytes["ssg_k_ext_lane_dyn<4>"] = alg_bytes.pop("ssg_k_ext_lane<136>")
            seed_k = sorted(k for k in kern if k.startswith(("ssg_k_smem2", "ssg_k_smem_heavy", "ssg_k_smem_quad", "ssg_k_smem_lane")))
            smk = " + ".join(seed_k) if seed_k else None
            seed_ms = sum(kern[k][0] for k in seed_k) / a.steps if seed_k else 0.0
            alg_seed = alg_bytes.pop("ssg_k_smem_quad")
            if smk:
                alg_bytes[smk] = alg_seed
                kern[smk] = (seed_ms * a.steps, a.steps)
            largest = name
            name, per_launch_ms = (smk, seed_ms) if smk else (name, per_launch_ms)
            alg = alg_bytes.get(name, 0.0)
            ach = alg / (per_launch_ms * 1e-3) / 1e9
            # HBM bytes per launch from the PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs,
            # tools/profile_round.sh; units calibrated on the random-gather probe): read from the committed summary when it was
            # taken on this workload with these kernels, otherwise null -- never a constant in this file
            traffic = None
            real = {"ssg_k_smem2_kt": "ssg_k_smem2<false, true>", "ssg_k_smem2_plain": "ssg_k_smem2<false, false>"}   # the launcher's names of the template instances (ssg_seed.cpp) -> rocprofv3's
            for tag in ("r06", "r05", "r04", "r02"):
                try:
                    pm = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json")))
                    if pm.get("pairs") == a.pairs and pm.get("read_len") == rl and abs(pm.get("ref_mbp", 0) - a.ref_mbp) < 1e-6 and seed_k and all(real.get(k, k) in pm.get("bytes_per_launch", {}) for k in seed_k):
                        traffic = sum(pm["bytes_per_launch"][real.get(k, k)] for k in seed_k)
                        break
                except Exception:
                    pass
            valu = {}; sw_lane_ops = None
            try:   # fraction of the measured int32 VALU issue peak (tools/dbg/valu_probe) from the committed SQ counter pass of this workload
                pq = json.load(open(next(f for f in (os.path.join(ROOT, "profiles", t + "_pmc_sq.json") for t in ("r06", "r05", "r04", "r02")) if os.path.exists(f))))
                valu = {k: round(v["valu_frac_of_probe_peak"], 3) for k, v in pq["kernels"].items() if k.startswith(("ssg_k_matesw", "ssg_k_msw_lane", "ssg_k_ext_lane", "ssg_k_smem")) and "valu_frac_of_probe_peak" in v and not k.endswith("_need")}
                # vector instructions of the SW kernels per step (committed counters of the same kernels' code: the ISA is pinned) x 64 lanes
                pmc_steps = max([1] + [v.get("launches", 1) for k, v in pq["kernels"].items() if k.startswith("ssg_k_matesw") and not k.startswith("ssg_k_matesw_need")])   # one mate-rescue launch per step of the counter run
                sw_lane_ops = None if not (a.pairs == 1000000 and rl == 150 and abs(a.ref_mbp - 3100.0) < 1e-6) else 64.0 * sum(v["SQ_INSTS_VALU_per_launch"] * v["launches"] / pmc_steps for k, v in pq["kernels"].items()
                                         if k.startswith(("ssg_k_matesw", "ssg_k_msw_lane", "ssg_k_ext_lane", "ssg_k_chain2aln", "ssg_k_reg2aln")) and not k.endswith("_need") and "SQ_INSTS_VALU_per_launch" in v)
            except Exception:
                pass
            sw_names = [k for k in kern if k.startswith(("ssg_k_matesw", "ssg_k_msw_lane", "ssg_k_ext_lane", "ssg_k_chain2aln", "ssg_k_reg2aln")) and not k.endswith("_need")]
            sw_ms = sum(kern[k][0] for k in sw_names) / a.steps
            out["roofline"] = {"bound": "hbm", "kernel": name, "largest_kernel": {"name": largest, "ms_per_launch": kern[largest][0] / kern[largest][1]}, "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                               "ms_per_launch": per_launch_ms,
                               "kernels_ms_per_step": {k: round(v[0] / a.steps, 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0]) if k != smk or len(seed_k) == 1},
                               "hbm_gbps_by_kernel": {k: round(alg_bytes[k] / (kern[k][0] / kern[k][1] * 1e-3) / 1e9, 1) for k in alg_bytes if k in kern},
                               # FM-index kernels read random 64-byte lines: measured MI355X gather peak (tools/dbg/gather_probe.cpp)
                               "random64B": {"kernel": smk, "peak_glines_per_s": 55.0, "peak_gbps": 3520.0,
                                             "achieved_glines_per_s": (2.0 * n_ext / (kern[smk][0] / kern[smk][1] * 1e-3) / 1e9) if smk else None,
                                             "frac": (2.0 * n_ext / (kern[smk][0] / kern[smk][1] * 1e-3) / 55e9) if smk else None},
                               "sw": {"cells_per_step": int(summary[3]) + int(summary[4]), "kernels": sorted(sw_names),
                                      "gcups": (int(summary[3]) + int(summary[4])) / (sw_ms * 1e-3) / 1e9 if sw_ms else None,
                                      "lane_ops_per_cell": (sw_lane_ops / (int(summary[3]) + int(summary[4]))) if sw_lane_ops and (int(summary[3]) + int(summary[4])) else None,   # SQ_INSTS_VALU x 64 of the SW kernels (committed counters, pinned ISA) over this run's DP cells
                                      "valu_frac": valu or None, "valu_frac_source": "profiles/r06_pmc_sq.json if present, else r05 / r04 / r02 (SQ_INSTS_VALU x 64 lanes / kernel time, over tools/dbg/valu_probe's add+max rate at 16 waves/CU)"}}
        # ---- parity gate ON THE TIMED CALL + CPU baseline: the step is run once more on the same device-resident inputs with its records
        # kept in HBM (identical inputs -> identical records; the summaries are compared), the records and samblaster's per-line decisions
        # are downloaded, and the oracle (scalar C restatement of bwa mem + samblaster) aligns the same pairs in the same upstream batches ----
        if a.cpu_sample > 0 and world == 1:   # rank 0 at N = 1 only
            import oracle_py
            import tempfile
            import common
            ns = min(a.cpu_sample, a.pairs)
            full = ns == a.pairs
            orc = oracle_py.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
            shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
            td_obj = tempfile.TemporaryDirectory(dir=shm)
            td = td_obj.name
            prefix = os.path.join(td, "ref.fa")
            tw = time.time()
            lib.index_save(idx, prefix)           # the five `bwa index` files of the device-built index
            oidx = orc.idx_load(prefix)           # ... loaded by the oracle: the index bytes cross the file format both ways
            t_files = time.time() - tw
            log('index files written and loaded by the oracle (%.1f s)' % t_files)
            hs_all = reads.cpu().numpy().reshape(-1)
            hs = hs_all[:2 * ns * rl]
            hoff = np.arange(2 * ns + 1, dtype=np.int64) * rl
            names = ["r%d" % (i // 2) for i in range(2 * ns)]
            cores = min(os.cpu_count() or 1, 128)   # the oracle's worker threads: every hardware thread the host shows (cgroup quotas may give less: tools/dbg/host_probe.py)
            hdr = "".join("@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in zip(ctg_names, lens))
            lf0 = (0, 0)
            try:
                lf0 = orc.lazyf(1)                     # count, on the first pairs, the mate-rescue alignments on which upstream's lazy-F pass could part from the textbook recurrence (oracle/orc_ksw.c)
            except Exception:
                pass
            nw = min(ns, 2000)                         # untimed: the worker threads' allocator heaps and the index pages they touch first
            orc.process_pairs(oidx, hs[:2 * nw * rl], hoff[:2 * nw + 1], names[:2 * nw], None, 0, "", cores)
            try:
                nlf = min(ns, 50000)
                orc.process_pairs(oidx, hs[:2 * nlf * rl], hoff[:2 * nlf + 1], names[:2 * nlf], None, 0, "", cores)
                lf1 = orc.lazyf(0)
                lazyf = {"pairs": nlf, "ksw_align2_calls": lf1[0] - lf0[0], "calls_whose_result_changes_without_E_from_F_raised_H": lf1[1] - lf0[1],
                         "what": "every mate-rescue alignment of these pairs a second time with E never opened from an H that F alone raised -- more than upstream's lazy-F pass leaves out -- "
                                 "0 differing calls = the textbook recurrence the kernels compute and upstream's agree on all of them"}
            except Exception as e:
                lazyf = {"error": repr(e)}
            tc = time.perf_counter()
            otext = ""
            nb_s = int(pb[ns - 1]) + 1
            for bi in range(nb_s):                     # one oracle call per upstream batch: the scope of the insert-size model
                lo = int(np.searchsorted(pb, bi, "left")); hi = min(ns, int(np.searchsorted(pb, bi, "right")))
                t_, _, _ = orc.process_pairs(oidx, hs[2 * lo * rl:2 * hi * rl], hoff[2 * lo:2 * hi + 1] - hoff[2 * lo], names[2 * lo:2 * hi], None, 2 * lo, "", cores)
                otext += t_
            tc = time.perf_counter() - tc
            log('oracle aligned %d pairs on %d threads in %.1f s' % (ns, cores, tc))
            om, od, os_ = common.oracle_streams(orc, otext, hdr)
            log('oracle samblaster done')
            # the device side: the timed call itself when the whole batch is checked, else the same entry point on the sample's pairs
            if full:
                s2, hrec = capi.hotpath_dev_ex(lib, idx, opt, a.pairs, rl, d_seq.data_ptr(), d_off.data_ptr(), d_pb.data_ptr(), n_batches, 0, keep=True)
                same_summary = bool(np.array_equal(np.asarray(s2), np.asarray(summary)))
            else:
                d_pbs = d_pb[:ns].contiguous()
                s2, hrec = capi.hotpath_dev_ex(lib, idx, opt, ns, rl, d_seq.data_ptr(), d_off.data_ptr(), d_pbs.data_ptr(), nb_s, 0, keep=True)
                same_summary = None
            res, bits, mate = capi.dev_records_download(lib, hrec, ns)
            gtext, _ = capi.sam_format(lib, idx, opt, res, names, hs, hoff, None, "")
            res.close(); capi.dev_records_free(lib, hrec)
            log('device records downloaded and printed')
            gm, gd, gs = common.sbl_streams_from_bits(gtext, bits, mate)
            log('streams rebuilt from the device decisions')
            ok_text = gtext == otext
            ok = bool(ok_text and gm == om and gd == od and gs == os_ and same_summary is not False)
            n_rec_diff = 0
            if not ok_text:
                ga, oa = gtext.split("\n"), otext.split("\n")
                n_rec_diff = sum(1 for x, y in zip(ga, oa) if x != y) + abs(len(ga) - len(oa))
                for x, y in list((x, y) for x, y in zip(ga, oa) if x != y)[:5]:
                    sys.stderr.write("[bench] PARITY MISMATCH\n  gpu    %s\n  oracle %s\n" % (x, y))
            out["parity"] = {"parity_checked_pairs": ns, "parity_ok": ok, "records_compared": otext.count("\n"), "records_differing": n_rec_diff,
                             "on_the_timed_call": full, "timed_call_summary_reproduced": same_summary,
                             "streams": {"sam_lines": [len(gm), len(om), gm == om], "discordant_lines": [len(gd), len(od), gd == od], "splitter_lines": [len(gs), len(os_), gs == os_]},
                             "dup_lines": int((bits & 1).sum()),
                             "what": ("ssg_hotpath_dev_ex -- the timed entry point, same device-resident inputs and upstream batches%s -- with its records kept in HBM: printed "
                                      "through ssg_sam_format they are the oracle's SAM text (all fields and tags), and their per-line duplicate / discordant / splitter bits and "
                                      "MC/MQ source lines reproduce the three streams of the oracle's samblaster line for line; index files written by ssg_index_save"
                                      % ("" if full else ", first %d pairs" % ns)),
                             "index_files_roundtrip_s": round(t_files, 1)}
            out["parity"]["ksw_align2_lazy_f"] = lazyf
            try:   # the one container of the hot path that was a shared deviation of oracle and kernels until round 6 (upstream mem_chain keeps the chains in klib's B-tree)
                ne = min(2 * ns, 400000)
                ex = orc.chain_exposure(oidx, hs[:int(hoff[ne])], hoff[:ne + 1], n_threads=cores)
                out["parity"]["chain_container"] = {"reads_checked": ex["reads"], "reads_with_more_than_9_chains": ex["gt9"], "reads_with_two_chains_at_one_position": ex["dup"],
                                                    "reads_with_both (can differ)": ex["gt9_and_dup"], "reads_whose_chains_differ_between_btree_and_array": ex["differ"],
                                                    "device_reads_rechained_on_the_btree (whole timed batch)": int(summary[11]) if len(summary) > 11 else None,
                                                    "what": "oracle/orc_mem.c runs every read of the sample through klib's B-tree (upstream's container, restated) and through the position-sorted array the kernels use for "
                                                            "unflagged reads; the kernels flag reads with more than 9 chains and two at one position and chain them again on the tree (csrc/k_chain.h ssg_k_chain_kb); "
                                                            "the oracle the records are compared with uses the tree"}
            except Exception as e:
                out["parity"]["chain_container"] = {"error": repr(e)}
            out["cpu_baseline"] = {"value": ns / tc, "unit": "pairs/s", "cores": cores, "host_cpu_quota": host_cpu_quota(), "hardware_threads": os.cpu_count(), "kind": "port",
                                   "sample": "%s of the timed batch (%d pairs) in its upstream batches, oracle/ (scalar C restatement of bwa mem PE) in process, %d threads, alignment only; after an untimed pass over %d pairs. "
                                             "The script-level baseline (`speedseq align -t <cores>` on the oracle's executables) is cpu_baseline.script" % ("all" if full else "the first pairs", ns, cores, nw)}
            try:   # oracle-independent validation of the same records (tests/validators.py): reference bases from the .pac just written
                import validators
                vtext = "\n".join(gtext.split("\n", 40001)[:40000]) + "\n"
                pac = validators.pac_contigs(prefix)
                v1, v2, v3 = validators.md_nm_consistency(vtext, pac), validators.as_from_cigar(vtext, pac), validators.mate_symmetry(vtext)
                out["parity"]["oracle_independent"] = {"md_nm_rebuilds_reference": {"records": v1[0], "bad": v1[1]},
                                                       "as_rescored_from_cigar": {"records": v2[0], "outside_[AS-10,AS]": v2[1]},
                                                       "mate_fields_mirror": {"checks": v3[0], "bad": v3[1]}}
                if v1[1] or v3[1] or v2[1] > v2[0] // 1000:
                    ok = False
                del pac
            except Exception as e:
                out["parity"]["oracle_independent"] = {"error": repr(e)}
            del gtext, otext, gm, gd, gs, om, od, os_
            if reads5 is not None:
                try:   # BASELINE.json configs[4]: 2x250, fragments 800 +- 150 -- the same entry point, timed and checked the same way on a smaller batch
                    n5, r5 = a.config5_pairs, 250
                    d5 = reads5.reshape(-1)
                    o5 = (torch.arange(2 * n5 + 1, device=dev, dtype=torch.int64) * r5).contiguous()
                    pb5, nb5 = bwa_batches(n5, r5, a.bwa_threads)
                    dpb5 = torch.from_numpy(pb5).to(dev)
                    run5 = lambda keep=False: capi.hotpath_dev_ex(lib, idx, opt, n5, r5, d5.data_ptr(), o5.data_ptr(), dpb5.data_ptr(), nb5, 0, keep=keep)
                    run5(); torch.cuda.synchronize()
                    t5 = time.perf_counter()
                    for _ in range(3):
                        s5, _ = run5()
                    torch.cuda.synchronize()
                    t5 = (time.perf_counter() - t5) / 3
                    ns5 = min(a.wide_pairs if a.wide_pairs > 0 else 5000, n5)
                    h5 = reads5[:2 * ns5].cpu().numpy().reshape(-1); ho5 = np.arange(2 * ns5 + 1, dtype=np.int64) * r5
                    nm5 = ["q%d" % (i // 2) for i in range(2 * ns5)]
                    ot5, _, _ = orc.process_pairs(oidx, h5, ho5, nm5, None, 0, "", cores)
                    _, hr5 = capi.hotpath_dev_ex(lib, idx, opt, ns5, r5, d5.data_ptr(), o5.data_ptr(), dpb5[:ns5].contiguous().data_ptr(), int(pb5[ns5 - 1]) + 1, 0, keep=True)
                    res5, b5, m5 = capi.dev_records_download(lib, hr5, ns5)
                    gt5, _ = capi.sam_format(lib, idx, opt, res5, nm5, h5, ho5, None, "")
                    res5.close(); capi.dev_records_free(lib, hr5)
                    same5 = gt5 == ot5 and common.sbl_streams_from_bits(gt5, b5, m5) == common.oracle_streams(orc, ot5, hdr)
                    out["config5"] = {"workload": "%d synthetic 2x250 bp pairs, fragments 800 +- 150 (BASELINE.json configs[4]), same reference and entry point" % n5,
                                      "ms_per_step": 1e3 * t5, "pairs_per_s": n5 / t5, "records": int(s5[0]), "rescues": int(s5[5]), "sw_cells": int(s5[3]) + int(s5[4]),
                                      "parity_checked_pairs": ns5, "parity_ok": bool(same5)}
                    if not same5:
                        ok = False
                    del d5, o5, dpb5
                except Exception as e:
                    out["config5"] = {"error": repr(e)}
                log('2x250 leg done')
                save_partial()
            if a.e2e:
                b = (lambda n: os.path.join(ROOT, "tests", "emu", n + "_emu")) if emu else (lambda n: os.path.join(ROOT, "bin", n))
                orc_exe = os.path.join(ROOT, "oracle", "orc_bwa")
                nse = min(20000, ns)
                log('parity gate done: %s' % ok)
                save_partial()
                try:
                    out["e2e"] = e2e_leg(a, td, prefix, reads, reads_e2e, rl, nse, orc_exe=orc_exe, b=b, reads_long=reads5, n_wide=a.wide_pairs)
                    if not out["e2e"].get("sample_streams_identical", True) or out["e2e"].get("parity_wide", {}).get("all_identical") is False:
                        ok = False
                except Exception as e:      # the plugin-path measurement must not take the headline down with it
                    out["e2e"] = {"error": repr(e)}
                log('plugin-path leg done')
                save_partial()
                try:
                    out["literal"] = literal_legs(a, td, prefix, rl, nse, b, orc_exe)
                    if out["literal"].get("sample_bams_equal_oracle") is False:
                        ok = False
                    if "cpu_script" in out["literal"]:
                        out["cpu_baseline"]["script"] = out["literal"].pop("cpu_script")
                    # the metric as SURVEY 8d words it, next to `value` (which this tier defines as the device step with inputs resident in HBM)
                    out["value_literal"] = {"value": out["literal"].get("value"), "unit": "pairs/s", "what": "FASTQ file -> three sorted BAMs + BAI through the reference's unmodified script, index load included (see `literal`)"}
                except Exception as e:
                    out["literal"] = {"error": repr(e)}
            log('script legs done')
            if a.dist_rehearsal and not emu and time.time() - _T0 < 360:
                try:   # the N > 1 code path (signature exchange, owner-side marking, range exchange of the records) on one rank, in a process of its own:
                       # what the exchange layer costs before a byte crosses xGMI.  No scaling curve exists until an 8-GPU box runs this script.
                    import subprocess
                    env = dict(os.environ, SSG_BENCH_FORCE_DIST="1", MASTER_PORT="29533")
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-e2e", "--cpu-sample", "0", "--no-profile",
                                        "--pairs", str(a.pairs), "--ref-mbp", str(a.ref_mbp), "--partial", a.partial + ".dist"], env=env, capture_output=True, text=True, timeout=240)
                    dj = json.loads(r.stdout.strip().split("\n")[-1])
                    out["dist_rehearsal"] = {"what": "SSG_BENCH_FORCE_DIST=1: the N > 1 step on one rank (nccl group of size 1)", "ms_per_step": dj["ms_per_step"],
                                             "overhead_ms_vs_single": dj["ms_per_step"] - out["ms_per_step"], "sorted_merge": dj["config"].get("sorted_merge")}
                except Exception as e:
                    out["dist_rehearsal"] = {"error": repr(e)[:300]}
                log('N > 1 rehearsal done')
            save_partial()
            td_obj.cleanup()
            if not ok:   # no timing counts without parity
                out["value"] = None
                out["invalid"] = "parity gate failed: GPU records differ from the oracle"
        if reads_ml is not None:
            try:   # `speedseq align` (unmodified script, fused hand-off) with bin/bwa driving all N devices; the other ranks wait at the barrier below
                import tempfile
                shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
                with tempfile.TemporaryDirectory(dir=shm) as td:
                    prefix = os.path.join(td, "ref.fa")
                    lib.index_save(idx, prefix)
                    fq = os.path.join(td, "reads.fq")
                    write_fastq(fq, reads_ml.numpy(), rl)
                    b = (lambda n: os.path.join(ROOT, "tests", "emu", n + "_emu")) if emu else (lambda n: os.path.join(ROOT, "bin", n))
                    cfg = ("export SSG_FUSED=1\nexport SSG_BWA_DEVICES=%d\nexport SSG_SORT_THREADS=%d\nexport SSG_FMT_THREADS=%d\nexport SSG_SORT_LOG=1\nexport SSG_SBL_LOG=1\nexport SSG_LOAD_LOG=1\n"
                           % (world, min(os.cpu_count() or 8, 256), min(os.cpu_count() or 8, 64)))
                    r = script_leg(td, "multi", prefix, fq, n_ml, a.script_threads, b("bwa"), b("samblaster"), b("sambamba"), config_extra=cfg, limit_s=180)
                    r.pop("out", None)
                    r["devices"] = world
                    r["what"] = "`speedseq align -t %d -p` (reference script, unmodified; SSG_FUSED=1) with bin/bwa mem driving %d devices: whole upstream batches per device, no collective; FASTQ -> three sorted BAMs + BAI, index load on every device included" % (a.script_threads, world)
                    out["literal_multi"] = r
                    log('script on %d devices: %s pairs in %s s' % (world, r.get('pairs'), r.get('wall_s')))
                    # ... and as `world` pipelines side by side, one per device (bin/speedseq-ranks): the same three BAMs, every stage N times
                    cfg_r = "export SSG_FUSED=1\nexport SSG_BWA_DEVICES=1\nexport SSG_SORT_LOG=1\nexport SSG_SBL_LOG=1\n"
                    r = script_leg(td, "ranks", prefix, fq, n_ml, a.script_threads, b("bwa"), b("samblaster"), b("sambamba"), sort_mem_gb=max(8, 40 // world), config_extra=cfg_r, limit_s=180, ranks=world,
                                   env_extra={"SSG_RANKS_KEEP_DEVICES": "1"} if emu else None)
                    r.pop("out", None)
                    r["devices"] = world
                    r["what"] = "bin/speedseq-ranks -n %d: the reference's script (unmodified) once per device, `bwa mem` aligning the upstream batches of its rank, the duplicate set sharded over the ranks' samblasters by signature (first seen wins in input order), both side streams in rank 0's, the sorts exchanging sorted runs and each placing its stretch of the genome in the one file; FASTQ -> the same three sorted BAMs + BAI" % world
                    out["literal_ranks"] = r
                    log('script as %d ranks: %s pairs in %s s' % (world, r.get('pairs'), r.get('wall_s')))
            except Exception as e:
                out["literal_multi"] = {"error": repr(e)[:400]}
        if saved_stdout is not None:
            C.CDLL(None).fflush(None); sys.stdout.flush(); os.dup2(saved_stdout, 1)
        print(json.dumps(out)); sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(2, 1)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
