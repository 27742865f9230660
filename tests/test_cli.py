"""Process-level drop-in boundary: the `bwa` and `samblaster` executables built on
libssgpu must emit the same three SAM streams as the CPU oracle's command line, byte for byte
(modulo the @PG lines).  CPU-side: the host-emulation build; `-m gpu`: the HIP build in bin/."""
import os
import subprocess

import pytest

import simreads
from common import EXAMPLE_FA, ROOT

ORC = os.path.join(ROOT, "oracle", "orc_bwa")
RG = "@RG\\tID:grp1\\tSM:s1\\tLB:lib1"


def _no_pg(text):
    return "\n".join(l for l in text.split("\n") if not l.startswith("@PG"))


def _run_pipeline(bwa, samblaster, ref, fq, d, tag, extra_bwa=()):
    spl, disc = os.path.join(d, tag + ".spl.sam"), os.path.join(d, tag + ".disc.sam")
    p1 = subprocess.run(bwa + ["mem", "-t", "4", "-p", "-R", RG] + list(extra_bwa) + [ref, fq], capture_output=True, check=True)
    p2 = subprocess.run(samblaster + ["--excludeDups", "--addMateTags", "--maxSplitCount", "2", "--minNonOverlap", "20",
                                      "--splitterFile", spl, "--discordantFile", disc], input=p1.stdout, capture_output=True, check=True)
    return p1.stdout.decode(), p2.stdout.decode(), open(spl).read(), open(disc).read()


def _check(bwa, samblaster, tmp_path, n_pairs, seed, extra_bwa=(), **sim):
    d = str(tmp_path)
    contigs = simreads.read_fasta(EXAMPLE_FA)
    fq = os.path.join(d, "reads.fq.gz")
    simreads.write_fastq(fq, simreads.simulate(contigs, n_pairs, seed=seed, **sim))
    got = _run_pipeline(bwa, samblaster, EXAMPLE_FA, fq, d, "got", extra_bwa)
    exp = _run_pipeline([ORC], [ORC, "samblaster"], EXAMPLE_FA, fq, d, "exp", extra_bwa)
    for g, e, what in zip(got, exp, ("bwa mem", "samblaster stdout", "splitters", "discordants")):
        assert _no_pg(g) == _no_pg(e), what
    assert "\t1" in got[1]   # sanity: SAM with flags


def test_cli_emu_matches_oracle(tmp_path, emu_lib):
    emu = os.path.join(ROOT, "tests", "emu")
    _check([os.path.join(emu, "bwa_emu")], [os.path.join(emu, "samblaster_emu")], tmp_path, 300, seed=21)


def test_cli_emu_reads_of_300_bases(tmp_path, emu_lib):
    # the reference's script takes reads of any length (bin/speedseq:196-200): 2x300 through the executables, text against the oracle's
    emu = os.path.join(ROOT, "tests", "emu")
    _check([os.path.join(emu, "bwa_emu")], [os.path.join(emu, "samblaster_emu")], tmp_path, 200, seed=23, read_len=300, ins_mean=900, ins_std=150)


def test_cli_emu_samblaster_small_chunks(tmp_path, emu_lib, monkeypatch):
    """device calls of 7 blocks each: the duplicate set in HBM must carry first-seen-wins across calls (and grow)"""
    monkeypatch.setenv("SSG_SBL_CHUNK", "7")
    monkeypatch.setenv("SSG_SBL_TABLE_SLOTS", "16")
    emu = os.path.join(ROOT, "tests", "emu")
    _check([os.path.join(emu, "bwa_emu")], [os.path.join(emu, "samblaster_emu")], tmp_path, 200, seed=24)


def test_cli_emu_insert_override(tmp_path, emu_lib):
    emu = os.path.join(ROOT, "tests", "emu")
    _check([os.path.join(emu, "bwa_emu")], [os.path.join(emu, "samblaster_emu")], tmp_path, 120, seed=22, extra_bwa=("-I", "400,50"))


def test_cli_error_behaviour(tmp_path, emu_lib):
    emu = os.path.join(ROOT, "tests", "emu", "bwa_emu")
    r = subprocess.run([emu, "mem", "-R", "RG\\tID:x", EXAMPLE_FA, "/dev/null"], capture_output=True)
    assert r.returncode != 0 and b"not started with @RG" in r.stderr
    r = subprocess.run([emu, "mem", "-p", EXAMPLE_FA + ".missing", "/dev/null"], capture_output=True)
    assert r.returncode != 0
    # the readers are already running when the index fails to load: more records than their channel holds must not keep the process alive
    big = tmp_path / "many.fq"
    big.write_text("".join("@r%d\nACGT\n+\nIIII\n" % i for i in range(200000)))
    r = subprocess.run([emu, "mem", "-p", EXAMPLE_FA + ".missing", str(big)], capture_output=True, timeout=60)
    assert r.returncode != 0 and b"fail to load the index" in r.stderr
    for env in ({"SSG_FASTQ_THREADS": "4", "SSG_FASTQ_PIECE": "20000"}, {}):
        r = subprocess.run([emu, "mem", EXAMPLE_FA + ".missing", str(big), str(big)], capture_output=True, timeout=60, env=dict(os.environ, **env))
        assert r.returncode != 0 and b"fail to load the index" in r.stderr
    # upstream letters this build does not take, and values that would need upstream's long-read path: an error that names the reason, never other output
    fq = tmp_path / "few.fq"
    simreads.write_fastq(str(fq), simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 50, seed=5))
    for opts, msg in ((["-a"], b"not supported"), (["-x", "pacbio"], b"not supported"), (["-W", "2"], b"chained-seed filter"), (["-A", "40"], b"DP cells")):
        r = subprocess.run([emu, "mem", "-p"] + opts + [EXAMPLE_FA, str(fq)], capture_output=True, timeout=120)
        assert r.returncode != 0 and msg in r.stderr, (opts, r.stderr[-300:])
        assert not any(l and l[0] != "@" for l in r.stdout.decode().split("\n")), opts


@pytest.mark.gpu
def test_cli_gpu_reads_of_300_bases(tmp_path, gpu_lib):
    _check([os.path.join(ROOT, "bin", "bwa")], [os.path.join(ROOT, "bin", "samblaster")], tmp_path, 2000, seed=23, read_len=300, ins_mean=900, ins_std=150)


@pytest.mark.gpu
def test_cli_gpu_matches_oracle(tmp_path, gpu_lib):
    _check([os.path.join(ROOT, "bin", "bwa")], [os.path.join(ROOT, "bin", "samblaster")], tmp_path, 6000, seed=23)


def _grammar_fastq(pairs, style):
    """the same reads in the shapes klib's kseq grammar admits: wrapped sequence / quality lines, FASTA records (no qualities),
    /1 /2 suffixes, comments, CRLF, lower case, blank lines between records"""
    out = []
    for k, (name, r1, r2) in enumerate(pairs):
        for end, r in ((1, r1), (2, r2)):
            s = "".join("ACGTN"[c] for c in r)
            q = "".join(chr(33 + (7 * i + k) % 40) for i in range(len(s)))
            nm = name + ("/%d" % end if style in ("suffix", "mixed") else "")
            com = " BC:Z:%04d\tXY:i:%d" % (k, end) if style in ("comment", "mixed") else ""
            eol = "\r\n" if style == "crlf" else "\n"
            if style == "lower":
                s = s.lower()
            if style == "fasta" or (style == "mixed" and k % 3 == 0):
                out.append(">" + nm + com + eol + s[:70] + eol + s[70:] + eol)
            elif style in ("wrapped", "mixed"):
                out.append("@" + nm + com + eol + s[:61] + eol + s[61:] + eol + "+" + nm + eol + q[:33] + eol + q[33:] + eol + (eol if k % 2 else ""))
            else:
                out.append("@" + nm + com + eol + s + eol + "+" + eol + q + eol)
    return "".join(out)


@pytest.mark.parametrize("style", ["plain", "wrapped", "fasta", "suffix", "comment", "crlf", "lower", "mixed"])
def test_cli_emu_fastq_grammar(tmp_path, emu_lib, style):
    """bin/bwa's reader (host/fastq.h) against the oracle's restatement of kseq_read + trim_readno, through `bwa mem [-C]`"""
    _grammar(tmp_path, os.path.join(ROOT, "tests", "emu", "bwa_emu"), style)


def _grammar(tmp_path, emu, style):
    pairs = simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 40, seed=77)
    fq = str(tmp_path / "g.fq")
    open(fq, "w", newline="").write(_grammar_fastq(pairs, style))
    extra = ["-C"] if style in ("comment", "mixed") else []
    got = subprocess.run([emu, "mem", "-t", "2", "-p"] + extra + [EXAMPLE_FA, fq], capture_output=True, check=True).stdout.decode()
    exp = subprocess.run([ORC, "mem", "-t", "2", "-p"] + extra + [EXAMPLE_FA, fq], capture_output=True, check=True).stdout.decode()
    assert _no_pg(got) == _no_pg(exp)
    assert got.count("\n") > 80


def test_cli_emu_two_files_and_truncation(tmp_path, emu_lib):
    _two_files(tmp_path, os.path.join(ROOT, "tests", "emu", "bwa_emu"))


def _two_files(tmp_path, emu):
    pairs = simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 60, seed=78)
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    for path, which in ((f1, 1), (f2, 2)):
        with open(path, "w") as f:
            for name, r1, r2 in pairs:
                r = r1 if which == 1 else r2
                f.write("@%s/%d\n%s\n+\n%s\n" % (name, which, "".join("ACGTN"[c] for c in r), "I" * len(r)))
    got = subprocess.run([emu, "mem", "-t", "2", EXAMPLE_FA, f1, f2], capture_output=True, check=True).stdout.decode()
    exp = subprocess.run([ORC, "mem", "-t", "2", EXAMPLE_FA, f1, f2], capture_output=True, check=True).stdout.decode()
    assert _no_pg(got) == _no_pg(exp)
    bad = str(tmp_path / "bad.fq")
    open(bad, "w").write(open(f1).read()[:-40])      # quality string cut short
    r = subprocess.run([emu, "mem", "-p", EXAMPLE_FA, bad], capture_output=True)
    assert r.returncode != 0


def test_cli_emu_many_batches_and_device_calls(tmp_path, emu_lib, monkeypatch):
    """upstream batches of ~130 pairs (one insert-size model each), grouped two or three to a device call, over a dozen calls: the
    batch assembly, pair ordinals and per-batch models of bin/bwa against the oracle's loop, through samblaster"""
    monkeypatch.setenv("SSG_BWA_CHUNK_BASES", "20000")
    monkeypatch.setenv("ORC_CHUNK_BASES", "20000")
    monkeypatch.setenv("SSG_BWA_CALL_PAIRS", "300")
    monkeypatch.setenv("SSG_SBL_CHUNK", "113")
    emu = os.path.join(ROOT, "tests", "emu")
    _check([os.path.join(emu, "bwa_emu")], [os.path.join(emu, "samblaster_emu")], tmp_path, 1700, seed=29)



@pytest.mark.parametrize("opts", [
    [],                                                                          # no --excludeDups (speedseq align -i), no mate tags
    ["--addMateTags"],
    ["--excludeDups", "--addMateTags", "--maxSplitCount", "1", "--minNonOverlap", "50"],
    ["--excludeDups", "--maxSplitCount", "3", "--minNonOverlap", "5"],
])
def test_cli_emu_samblaster_option_sets(tmp_path, emu_lib, opts):
    """the option sets `speedseq align` can produce (-i, -c, -m: bin/speedseq:228-243) and samblaster's own defaults, vs the oracle"""
    _option_sets(tmp_path, os.path.join(ROOT, "tests", "emu", "samblaster_emu"), opts)


def _option_sets(tmp_path, samblaster_exe, opts):
    fq = str(tmp_path / "r.fq.gz")
    simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 500, seed=51, chim_frac=0.05, disc_frac=0.05, dup_frac=0.1))
    sam = subprocess.run([ORC, "mem", "-t", "4", "-p", EXAMPLE_FA, fq], capture_output=True, check=True).stdout
    res = []
    for tag, exe in (("got", [samblaster_exe]), ("exp", [ORC, "samblaster"])):
        spl, disc = str(tmp_path / (tag + ".spl")), str(tmp_path / (tag + ".disc"))
        p = subprocess.run(exe + opts + ["--splitterFile", spl, "--discordantFile", disc], input=sam, capture_output=True, check=True)
        res.append((_no_pg(p.stdout.decode()), _no_pg(open(spl).read()), _no_pg(open(disc).read())))
    assert res[0] == res[1]
    assert res[0][2].count("\n") > 5 and (res[0][1].count("\n") > 5 or "1" in opts[opts.index("--maxSplitCount") + 1:][:1])   # one piece per read can never be a split


def test_cli_emu_bgzf_fastq_input(tmp_path, emu_lib):
    """blocked gzip (bgzip) FASTQ: members located by their headers and inflated by several threads (fastq.h bgzf_loop) -- same SAM as the plain file"""
    import struct
    import zlib
    pairs = simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 300, seed=21)
    fq = str(tmp_path / "r.fq")
    simreads.write_fastq(fq, pairs)
    data = open(fq, "rb").read()
    out = b""
    for o in range(0, len(data), 7001):                      # many small members, cut in the middle of records
        c = data[o:o + 7001]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = co.compress(c) + co.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(d) + 25) + d + struct.pack("<II", zlib.crc32(c), len(c))
    out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    open(fq + ".gz", "wb").write(out)
    exe = os.path.join(ROOT, "tests", "emu", "bwa_emu")
    a = subprocess.run([exe, "mem", "-p", EXAMPLE_FA, fq], capture_output=True, env=dict(os.environ, SSG_BGZF_THREADS="3"))
    b = subprocess.run([exe, "mem", "-p", EXAMPLE_FA, fq + ".gz"], capture_output=True, env=dict(os.environ, SSG_BGZF_THREADS="3"))
    strip = lambda x: [l for l in x.split(b"\n") if not l.startswith(b"@PG")]
    assert a.returncode == 0 and b.returncode == 0 and strip(a.stdout) == strip(b.stdout) and len(a.stdout) > 100000


def test_cli_emu_deferred_dense_suffix_array(tmp_path, emu_lib):
    """bin/bwa makes the denser suffix-array copy only once the input has proved long (SSG_BWA_DENSIFY_AFTER pairs per device; 0 = when
    the index is loaded): the SAM text is the same whether seeds were located through the file's samples, the dense copy, or first one
    then the other in the middle of a run"""
    exe = os.path.join(ROOT, "tests", "emu", "bwa_emu")
    fq = str(tmp_path / "r.fq")
    simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 600, seed=91))
    outs, errs = [], []
    for after in ("0", "250", "1000000"):
        env = dict(os.environ, SSG_BWA_DENSIFY_AFTER=after, SSG_BWA_CHUNK_BASES="20000", SSG_BWA_CALL_PAIRS="100", SSG_EMU_DEVICES="2")
        r = subprocess.run([exe, "mem", "-t", "2", "-p", EXAMPLE_FA, fq], capture_output=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(_no_pg(r.stdout.decode()))
        errs.append(r.stderr.decode())
    assert outs[0].count("\n") > 1200 and outs[1] == outs[0] and outs[2] == outs[0]
    assert "denser suffix-array copy made after" in errs[1] and "denser suffix-array" not in errs[0] and "denser suffix-array" not in errs[2]


def test_cli_emu_several_calls_in_flight(tmp_path, emu_lib):
    """SSG_BWA_INFLIGHT = 2 or 3 worker threads per device, each on a lane of its own (ssg_set_lane), with the suffix-array densification
    falling between their calls: the SAM text is that of one call at a time"""
    exe = os.path.join(ROOT, "tests", "emu", "bwa_emu")
    fq = str(tmp_path / "r.fq")
    simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 900, seed=93))
    outs = []
    for inflight, devices, after, fmts in (("1", "1", "0", "1"), ("2", "2", "300", "3"), ("3", "1", "200", "2")):   # ... and several formatter threads behind them
        env = dict(os.environ, SSG_BWA_INFLIGHT=inflight, SSG_EMU_DEVICES=devices, SSG_BWA_DENSIFY_AFTER=after, SSG_BWA_FORMATTERS=fmts, SSG_BWA_CHUNK_BASES="20000", SSG_BWA_CALL_PAIRS="90")
        r = subprocess.run([exe, "mem", "-t", "2", "-p", EXAMPLE_FA, fq], capture_output=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(_no_pg(r.stdout.decode()))
    assert outs[0].count("\n") > 1800 and outs[1] == outs[0] and outs[2] == outs[0]


BWA_OPTION_SETS = [
    ("-M",), ("-Y",), ("-M", "-Y"), ("-S",), ("-P",), ("-S", "-P"),
    ("-k", "25"), ("-w", "40", "-d", "50"), ("-A", "2"), ("-A", "2", "-B", "5", "-O", "7,9", "-E", "2,1"), ("-L", "3,8", "-U", "9"),
    ("-T", "45"), ("-c", "50", "-D", "0.3"), ("-r", "1.0", "-y", "10"), ("-m", "5", "-W", "6"), ("-h", "2", "-X", "0.3"), ("-K", "20000"),
    ("-Q", "20", "-s", "5", "-G", "500", "-N", "3"),
    ("-k", "8"),   # seeds shorter than the patterns of the index's table of short-pattern intervals: the seeding kernel uses the table's levels below 8 (ssg_seed.cpp kt_k)
]


@pytest.mark.parametrize("opts", BWA_OPTION_SETS, ids=lambda o: "".join(o))
def test_cli_emu_bwa_mem_option_sets(tmp_path, emu_lib, opts):
    """upstream main_mem's option letters (fastmap.c, 0.7.12): each set through the product's `bwa mem` and the oracle's, SAM byte for byte.
    -M / -Y change how split hits are printed, -S / -P drop mate rescue / pairing, the rest are the scoring and heuristic knobs of mem_opt_t
    (the `-A` forms also exercise upstream's update_a scaling of the penalties left alone)."""
    _bwa_options(tmp_path, [os.path.join(ROOT, "tests", "emu", "bwa_emu")], opts, 400)


def _bwa_options(tmp_path, bwa, opts, n_pairs):
    d = str(tmp_path)
    fq = os.path.join(d, "reads.fq")
    simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), n_pairs, seed=77))
    run = lambda exe: _no_pg(subprocess.run(exe + ["mem", "-t", "2", "-p", "-R", RG] + list(opts) + [EXAMPLE_FA, fq], capture_output=True, check=True).stdout.decode())
    got, exp = run(bwa), run([ORC])
    assert exp.count("\n") > 2 * n_pairs
    if "-M" in opts:
        assert any(int(l.split("\t")[1]) & 0x100 for l in exp.split("\n") if l and l[0] != "@"), "no split hit in the sample: -M untested"
    assert got == exp


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [("-M", "-Y"), ("-S", "-P"), ("-A", "2", "-B", "5", "-O", "7,9", "-E", "2,1"), ("-k", "25", "-c", "50", "-D", "0.3", "-r", "1.0"), ("-k", "8")], ids=lambda o: "".join(o))
def test_cli_gpu_bwa_mem_option_sets(tmp_path, gpu_lib, opts):
    _bwa_options(tmp_path, [os.path.join(ROOT, "bin", "bwa")], opts, 20000)


def _single_end(tmp_path, bwa, n_pairs, seed, opts=(), env=None, **kw):
    """one FASTQ without -p: upstream aligns every read on its own (mem_process_seqs without MEM_F_PE: primary marking by read ordinal, no
    insert-size model, no rescue, no mate fields)"""
    d = str(tmp_path)
    fq = os.path.join(d, "se_%d.fq" % seed)
    pairs = simreads.simulate(simreads.read_fasta(EXAMPLE_FA), n_pairs, seed=seed, **kw)
    with open(fq, "w") as f:   # both ends as independent reads with names of their own, and one more so that the count is odd
        k = 0
        for name, a, b in pairs:
            for s in (a, b):
                f.write("@%s_%d\n%s\n+\n%s\n" % (name, k, "".join("ACGTN"[c] for c in s), "I" * len(s)))
                k += 1
        f.write("@tail\n%s\n+\n%s\n" % ("".join("ACGTN"[c] for c in pairs[0][1]), "I" * len(pairs[0][1])))
    run = lambda exe, e=None: _no_pg(subprocess.run(exe + ["mem", "-t", "2", "-R", RG] + list(opts) + [EXAMPLE_FA, fq], capture_output=True, check=True, env=e).stdout.decode())
    exp = run([ORC])
    assert exp.count("\n") > 2 * n_pairs and not any(int(l.split("\t")[1]) & 0xc1 for l in exp.split("\n") if l and l[0] != "@")
    assert run(bwa, dict(os.environ, **(env or {}))) == exp


@pytest.mark.parametrize("seed,opts,env,kw", [(61, (), {}, {}), (62, ("-M", "-Y"), {"SSG_BWA_CALL_PAIRS": "97"}, {}), (63, ("-k", "25", "-T", "40"), {}, {"read_len": 250, "ins_mean": 800, "ins_std": 150})],
                         ids=["default", "MY_many_calls", "2x250_k25"])
def test_cli_emu_single_end(tmp_path, emu_lib, seed, opts, env, kw):
    _single_end(tmp_path, [os.path.join(ROOT, "tests", "emu", "bwa_emu")], 300, seed, opts, env, **kw)


@pytest.mark.gpu
def test_cli_gpu_single_end(tmp_path, gpu_lib):
    _single_end(tmp_path, [os.path.join(ROOT, "bin", "bwa")], 20000, 64)
