"""Rank mode (bin/speedseq-ranks, speedseq_amd/host/ranks.h): N pipelines of the reference's unmodified script side by side -- one per GPU on a
multi-GPU node, here N processes on the host emulation -- must end in the SAME three sorted BAMs as one pipeline: upstream's batches dealt
round-robin (insert-size models unchanged), ONE duplicate set (sharded over the ranks by signature) asked in batch order, side-stream lines leaving rank 0 in batch order, equal sort
keys in input order across ranks, every rank writing one stretch of the genome."""
import os
import shutil
import subprocess

import pytest

import simreads
from common import EXAMPLE_FA, ROOT
from test_speedseq_script import REF_SCRIPT, SAMTOOLS, EMU, _need_tools, _view
from test_sambamba import _parse_bai

RG = "@RG\\tID:NA12878\\tSM:NA12878\\tLB:lib1"


def _setup(d, config_extra, exe=None):
    exe = exe or (lambda name: os.path.join(EMU, name + "_emu"))
    os.makedirs(d)
    bindir = os.path.join(d, "bin")
    os.makedirs(bindir)
    for name in ("bwa", "samblaster"):
        with open(os.path.join(bindir, name), "w") as f:
            f.write("#!/bin/sh\nexec %s \"$@\"\n" % exe(name))
        os.chmod(os.path.join(bindir, name), 0o755)
    os.symlink(shutil.which("mawk"), os.path.join(bindir, "gawk"))
    cfg = os.path.join(d, "speedseq.config")
    with open(cfg, "w") as f:
        f.write("BWA=%s/bwa\nSAMBLASTER=%s/samblaster\nSAMBAMBA=%s\nPARALLEL=%s/bin/parallel\nexport SSG_FUSED=1\n%s" % (bindir, bindir, exe("sambamba"), ROOT, config_extra))
    ref = os.path.join(d, "ref.fa")
    shutil.copy(EXAMPLE_FA, ref)
    for ext in ("amb", "ann", "bwt", "pac", "sa"):
        shutil.copy(EXAMPLE_FA + "." + ext, ref + "." + ext)
    return cfg, ref, dict(os.environ, PATH="%s:%s" % (bindir, os.environ["PATH"]), SSG_BWA_CHUNK_BASES="40000", SSG_RANKS_KEEP_DEVICES="1", SSG_RDV_TIMEOUT="120")


def _records(bam):
    """the records' bytes in file order (the header aside): equal keys must be in the same order too"""
    import gzip
    import struct
    raw = gzip.open(bam, "rb").read()
    l_text, = struct.unpack_from("<i", raw, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, o)
    o += 4
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", raw, o)
        o += 4 + l + 4
    return raw[o:]


@pytest.mark.parametrize("world,extra,chunk,kw", [(2, "", "40000", {}), (3, "export SSG_SORT_CHUNK_BYTES=300000\n", "40000", {}), (4, "", "150000", {}),
                                                  (2, "export SSG_RANKS_SPLIT=0\n", "60000", {"read_len": 250, "ins_mean": 800, "ins_std": 150}),
                                                  (3, "", "40000", {"gz_two_files": True}), (8, "", "30000", {}),
                                                  (3, "export SSG_RANKS_SCAN_SLICE=3000\nexport SSG_RANKS_SCAN_THREADS=3\nexport SSG_RANKS_SHARD=0\nexport SSG_RANKS_JOIN=0\n", "40000", {})],
                         ids=["two_ranks", "three_ranks_spilling", "four_ranks_three_batches", "two_ranks_2x250_everyone_parses", "three_ranks_two_gz_files", "eight_ranks",
                              "three_ranks_tiny_scan_slices_one_table_launcher_join"])
def test_ranks_emulated_equal_one_pipeline(tmp_path, emu_lib, world, extra, chunk, kw):
    _ranks_equal_one(tmp_path, world, extra, chunk, None, 2500 if not kw else 1200, **kw)


@pytest.mark.parametrize("world,extra,chunk,expect", [(2, "export SSG_RANKS_XCHG=sock\n", "40000", "exchanged over UNIX sockets"), (4, "export SSG_RANKS_XCHG=sock\n", "40000", "exchanged over UNIX sockets"),
                                                      (3, "export SSG_RANKS_XCHG=sock\nexport SSG_SORT_CHUNK_BYTES=300000\n", "40000", "the ranks exchange through files")],
                         ids=["two_ranks_collective", "four_ranks_collective", "three_ranks_collective_a_rank_spills"])
def test_ranks_emulated_collective_exchange_equals_one_pipeline(tmp_path, emu_lib, world, extra, chunk, expect):
    """the sorts' exchange as ONE all-to-all of (ordinal, record) blocks (speedseq_amd/host/xchg.h; on a node of MI355X: grouped ncclSend / ncclRecv, csrc/ssg_coll.cpp;
    here the same code over the socket transport, the stand-in where there is no RCCL) instead of sorted runs in files -- the same three BAMs, byte for byte; and when a rank
    has spilled its input, every rank agrees -- by the same exchange -- to go through the files"""
    err = _ranks_equal_one(tmp_path, world, extra + "export SSG_DEBUG=1\n", chunk, None, 2500)
    assert expect in err, err[-1500:]


@pytest.mark.gpu
def test_ranks_gpu_rccl_exchange_one_rank_loopback(tmp_path, gpu_lib):
    """the RCCL transport itself on the one MI355X of the test box: a communicator of ONE rank, the all-to-all to itself through HBM (ncclSend / ncclRecv to the own rank),
    blocks and sizes back unchanged -- what a node with a GPU per rank runs between N of them (N > 1 ranks cannot share a device under RCCL: the N-rank run of this transport
    is the driver's multi-GPU bench)"""
    import ctypes as C
    import numpy as np
    lib = gpu_lib.l
    assert lib.ssg_coll_available() == 1
    h = C.c_void_p()
    rc = lib.ssg_coll_init(0, 1, str(tmp_path).encode(), C.byref(h))
    assert rc == 0, lib.ssg_last_error()
    for n in (0, 1, 1000, 50_000_000):
        src = np.random.default_rng(n).integers(0, 256, n, dtype=np.uint8); dst = np.zeros(n, dtype=np.uint8)
        sp = (C.c_void_p * 1)(src.ctypes.data); rp = (C.c_void_p * 1)(dst.ctypes.data); nb = (C.c_uint64 * 1)(n)
        assert lib.ssg_coll_alltoallv(h, sp, nb, rp, nb) == 0, lib.ssg_last_error()
        assert np.array_equal(src, dst)
    a = np.arange(3, dtype=np.uint64) + 7; b = np.zeros(3, dtype=np.uint64)
    assert lib.ssg_coll_alltoall_u64(h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), 3) == 0 and np.array_equal(a, b)
    lib.ssg_coll_destroy(h)


@pytest.mark.gpu
def test_ranks_gpu_two_pipelines_on_the_one_device(tmp_path, gpu_lib):
    """the same on the MI355X: two pipelines side by side, both on the one GPU of the test box (SSG_RANKS_KEEP_DEVICES=1)"""
    _ranks_equal_one(tmp_path, 2, "", "300000", lambda name: os.path.join(ROOT, "bin", name), 6000)


def _ranks_equal_one(tmp_path, world, extra, chunk, exe, n_pairs, **kw):
    _need_tools()
    two = kw.pop("gz_two_files", False)
    fq = str(tmp_path / ("reads_1.fq.gz" if two else "reads.fq"))
    fq2 = str(tmp_path / "reads_2.fq.gz") if two else None
    simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), n_pairs, seed=13, **kw), interleaved=not two, path2=fq2)
    tail = (["-p"] if not two else []) + ["-R", RG]
    cfg, ref, env = _setup(str(tmp_path / "one"), "", exe)
    env["SSG_BWA_CHUNK_BASES"] = chunk                  # x -t 2: 267 or 1000 pairs per upstream batch (the last case: fewer batches than ranks)
    one = str(tmp_path / "one" / "out")
    r = subprocess.run(["bash", REF_SCRIPT, "align", "-K", cfg, "-o", one, "-M", "3", "-t", "2"] + tail + [ref, fq] + ([fq2] if two else []), cwd=str(tmp_path / "one"), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    cfg, ref, env = _setup(str(tmp_path / "many"), extra, exe)
    env["SSG_BWA_CHUNK_BASES"] = chunk
    many = str(tmp_path / "many" / "out")
    r = subprocess.run([os.path.join(ROOT, "bin", "speedseq-ranks"), "-n", str(world), "--script", REF_SCRIPT, "--", "align", "-K", cfg, "-o", many, "-M", "3", "-t", "2"] + tail + [ref, fq] + ([fq2] if two else []),
                       cwd=str(tmp_path / "many"), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert ("stretches copied by the launcher" if "SSG_RANKS_JOIN=0" in extra else "stretches placed by the ranks") in r.stderr, r.stderr[-500:]   # every sort put its stretch into the one file itself (sambamba_main.cpp), the launcher only indexed
    for suffix in (".bam", ".splitters.bam", ".discordants.bam"):
        assert _view(many + suffix) == _view(one + suffix), suffix
        assert _records(many + suffix) == _records(one + suffix), suffix
        assert os.path.exists(many + suffix + ".bai")
    os.rename(many + ".bam.bai", many + ".mine.bai")
    subprocess.run([SAMTOOLS, "index", many + ".bam"], check=True)
    assert _parse_bai(many + ".mine.bai") == _parse_bai(many + ".bam.bai")
    assert int(subprocess.check_output([SAMTOOLS, "view", "-c", "-f", "1024", many + ".bam"])) > 50
    left = [f for f in os.listdir(str(tmp_path / "many")) if ".rank" in f]
    assert left == [], left
    return r.stderr


def test_ranks_need_the_fused_hand_off(tmp_path, emu_lib):
    """without `export SSG_FUSED=1` the stages cannot carry batch ordinals: `bwa mem` says so and the launcher reports the failed rank"""
    _need_tools()
    fq = str(tmp_path / "reads.fq")
    simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 200, seed=14))
    cfg, ref, env = _setup(str(tmp_path / "many"), "")
    text = open(cfg).read().replace("export SSG_FUSED=1\n", "")
    open(cfg, "w").write(text)
    r = subprocess.run([os.path.join(ROOT, "bin", "speedseq-ranks"), "-n", "2", "--script", REF_SCRIPT, "--", "align", "-K", cfg, "-o", str(tmp_path / "many" / "out"), "-M", "3", "-t", "2", "-p", "-R", RG, ref, fq],
                       cwd=str(tmp_path / "many"), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "rank mode needs the fused hand-off" in r.stderr and "failed" in r.stderr


def _bwa_ranks(tmp_path, world, fqs, env_extra=None, expect_fail=False):
    """`bwa mem` alone as `world` ranks (fused frames, as in rank mode): (return codes, stderr texts, the batches file of the run or None)"""
    rdv = str(tmp_path / "rdv")
    shutil.rmtree(rdv, ignore_errors=True)
    os.makedirs(rdv)
    env = dict(os.environ, SSG_FUSED="1", SSG_WORLD=str(world), SSG_RDV=rdv, SSG_RDV_TIMEOUT="60", SSG_BWA_CHUNK_BASES="40000", SSG_FUSED_SHM="0", **(env_extra or {}))
    procs = [subprocess.Popen([os.path.join(EMU, "bwa_emu"), "mem", "-t", "2"] + ([] if len(fqs) == 2 else ["-p"]) + ["-R", RG, EXAMPLE_FA] + fqs, env=dict(env, SSG_RANK=str(r)),
                              stdout=open(str(tmp_path / ("frames.%d" % r)), "wb"), stderr=subprocess.PIPE) for r in range(world)]
    errs = [p.communicate()[1].decode() for p in procs]
    b = os.path.join(rdv, "batches")
    return [p.returncode for p in procs], errs, (open(b, "rb").read() if os.path.exists(b) else None)


def test_ranks_read_only_their_share_of_plain_fastq(tmp_path, emu_lib):
    """plain regular files: rank 0 scans the input for upstream's batches and publishes their byte ranges (SSG_RDV/batches), every rank parses the
    ranges of its batches only; compressed input: rank 0 alone inflates it and hands every batch on as a file of its own; either way the frames
    are the same bytes as when every rank reads and parses everything (SSG_RANKS_SPLIT=0)"""
    import gzip
    pairs = simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 2000, seed=17)
    one = str(tmp_path / "il.fq")
    simreads.write_fastq(one, pairs)
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    simreads.write_fastq(f1, pairs, interleaved=False, path2=f2)
    gz = str(tmp_path / "il.fq.gz")
    with open(one, "rb") as fi, gzip.open(gz, "wb", compresslevel=1) as fo:
        fo.write(fi.read())
    frames = {}
    for tag, fqs, env in (("split", [one], {}), ("all", [one], {"SSG_RANKS_SPLIT": "0"}), ("gz", [gz], {}), ("gz_all", [gz], {"SSG_RANKS_SPLIT": "0"}), ("two", [f1, f2], {}), ("two_all", [f1, f2], {"SSG_RANKS_SPLIT": "0"})):
        rcs, errs, batches = _bwa_ranks(tmp_path, 3, fqs, env)
        assert rcs == [0, 0, 0], errs
        assert (batches is not None) == (tag in ("split", "two", "gz")), tag       # gz: rank 0 inflates and scans, the batches travel as files of their own
        assert [f for f in os.listdir(str(tmp_path / "rdv")) if f.startswith("fq.")] == [], tag
        if batches is not None:
            assert len(batches) % 40 == 0 and len(batches) // 40 >= 6
        frames[tag] = [open(str(tmp_path / ("frames.%d" % r)), "rb").read() for r in range(3)]
    def body(fr):     # the frames without the header frame (its @PG line carries the command line)
        out = []
        for b in fr:
            o = 8
            import struct
            t, z, l = struct.unpack_from("<IIQ", b, o)
            assert t == 1
            out.append(b[o + 16 + l:])
        return out
    assert body(frames["split"]) == body(frames["all"]) == body(frames["gz"]) == body(frames["gz_all"])
    assert body(frames["two"]) == body(frames["two_all"]) == body(frames["split"])


def test_ranks_split_refuses_fastq_that_is_not_four_lines_a_record(tmp_path, emu_lib):
    pairs = simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 300, seed=18)
    fq = str(tmp_path / "wrapped.fq")
    with open(fq, "w") as f:
        for name, a, b in pairs:
            for s in (a, b):
                t = "".join("ACGTN"[c] for c in s)
                f.write("@%s\n%s\n%s\n+\n%s\n" % (name, t[:70], t[70:], "I" * len(t)))     # the sequence on two lines: kseq reads it, the scanner does not
    rcs, errs, _ = _bwa_ranks(tmp_path, 2, [fq])
    assert all(rc != 0 for rc in rcs) and "SSG_RANKS_SPLIT=0" in errs[0], errs
    rcs, errs, _ = _bwa_ranks(tmp_path, 2, [fq], {"SSG_RANKS_SPLIT": "0"})
    assert rcs == [0, 0], errs


def test_ranks_stop_together_when_one_gives_up(tmp_path, emu_lib):
    """the reference's script does not stop when a stage of its pipeline fails, and ranks wait for one another: a stage that gives up leaves a mark
    in the rendezvous directory and every wait looks for one.  Here the scanner meets a wrapped record in the middle of the input, after all
    ranks have started: the run ends at once, with the reason, instead of waiting for SSG_RDV_TIMEOUT"""
    _need_tools()
    pairs = simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 1500, seed=19)
    fq = str(tmp_path / "half_wrapped.fq")
    with open(fq, "w") as f:
        for k, (name, a, b) in enumerate(pairs):
            for s_ in (a, b):
                t = "".join("ACGTN"[c] for c in s_)
                f.write("@%s\n%s\n+\n%s\n" % (name, t if k < 750 else t[:70] + "\n" + t[70:], "I" * len(t)))
    cfg, ref, env = _setup(str(tmp_path / "many"), "")
    env["SSG_RDV_TIMEOUT"] = "600"
    r = subprocess.run([os.path.join(ROOT, "bin", "speedseq-ranks"), "-n", "3", "--script", REF_SCRIPT, "--", "align", "-K", cfg, "-o", str(tmp_path / "many" / "out"), "-M", "3", "-t", "2", "-p", "-R", RG, ref, fq],
                       cwd=str(tmp_path / "many"), env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "SSG_RANKS_SPLIT=0" in r.stderr
    assert not os.path.exists(str(tmp_path / "many" / "out.bam"))
