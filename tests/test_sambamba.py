"""Row f1: the native `sambamba` (speedseq_amd/host/sambamba_main.cpp + bamio.h) against the reference's own
samtools 1.3.1 built into oracle/_ref/ (the tool the round-1 shim wrapped): SAM->BAM records, coordinate order incl. ties,
BGZF readability, BAI content, merge.  CPU side: the host-emulation build (device sort emulated); -m gpu: bin/sambamba."""
import os
import struct
import subprocess

import pytest

import simreads
from common import EXAMPLE_FA, ROOT

SAMTOOLS = os.path.join(ROOT, "oracle", "_ref", "samtools")
ORC = os.path.join(ROOT, "oracle", "orc_bwa")


def _sam(tmp_path, n_pairs=1200, seed=31, rg="g"):
    """name-grouped SAM of the oracle's bwa mem | samblaster (duplicates, supplementary lines, unmapped reads, tags)"""
    if not os.path.exists(SAMTOOLS):
        pytest.skip("oracle/_ref/samtools not built")
    fq = str(tmp_path / "r.fq.gz")
    simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), n_pairs, seed=seed))
    p1 = subprocess.run([ORC, "mem", "-t", "4", "-p", "-R", "@RG\\tID:%s\\tSM:s" % rg, EXAMPLE_FA, fq], capture_output=True, check=True)
    p2 = subprocess.run([ORC, "samblaster", "--addMateTags"], input=p1.stdout, capture_output=True, check=True)
    sam = str(tmp_path / "in.sam")
    with open(sam, "wb") as f:
        f.write(p2.stdout)
        # aux types beyond what bwa prints: negative / wide integers, float, hex, arrays, char
        f.write(b"xt1\t4\t*\t0\t0\t*\t*\t0\t0\tACGTNacgtn\t*\tXa:A:Q\tXb:i:-7\tXc:i:-300\tXd:i:-70000\tXe:i:255\tXf:i:256\tXg:i:70000\tXh:f:1.5\tXi:H:1AE3\tXj:B:c,-1,2\tXk:B:S,1,65535\tXl:B:f,0.5,2\tXm:Z:x y\n")
    return sam


def _view(bam, region=None):
    cmd = [SAMTOOLS, "view", "-h", bam] + ([region] if region else [])
    return subprocess.check_output(cmd, text=True)


def _parse_bai(path):
    b = open(path, "rb").read()
    assert b[:4] == b"BAI\1"
    n_ref, = struct.unpack_from("<i", b, 4)
    o, refs = 8, []
    for _ in range(n_ref):
        n_bin, = struct.unpack_from("<i", b, o); o += 4
        bins = {}
        for _ in range(n_bin):
            bid, n_chunk = struct.unpack_from("<Ii", b, o); o += 8
            bins[bid] = [struct.unpack_from("<QQ", b, o + 16 * k) for k in range(n_chunk)]; o += 16 * n_chunk
        n_intv, = struct.unpack_from("<i", b, o); o += 4
        lin = list(struct.unpack_from("<%dQ" % n_intv, b, o)); o += 8 * n_intv
        refs.append((bins, lin))
    n_no_coor = struct.unpack_from("<Q", b, o)[0] if o + 8 <= len(b) else None
    return refs, n_no_coor


def _check(sambamba, tmp_path, monkeypatch):
    sam = _sam(tmp_path)
    d = str(tmp_path)
    # view: SAM -> BAM, uncompressed and compressed
    for lvl, tag in ((["-l", "0"], "u"), ([], "c")):
        with open(sam, "rb") as fi, open("%s/mine_%s.bam" % (d, tag), "wb") as fo:
            subprocess.run(sambamba + ["view", "-S", "-f", "bam"] + lvl + ["/dev/stdin"], stdin=fi, stdout=fo, check=True)
    subprocess.run([SAMTOOLS, "view", "-b", "-u", "-o", d + "/ref_u.bam", sam], check=True)
    assert _view(d + "/mine_u.bam") == _view(d + "/ref_u.bam") == _view(d + "/mine_c.bam")
    assert subprocess.check_output(sambamba + ["view", "-H", d + "/mine_c.bam"], text=True) == subprocess.check_output([SAMTOOLS, "view", "-H", d + "/ref_u.bam"], text=True)
    # sort: identical record order (ties keep input order), SO:coordinate header
    subprocess.run(sambamba + ["sort", "-t", "3", "-m", "1G", "--tmpdir=" + d + "/tmp1", "-o", d + "/mine_s.bam", d + "/mine_u.bam"], check=True)
    subprocess.run([SAMTOOLS, "sort", "-o", d + "/ref_s.bam", d + "/ref_u.bam"], check=True)
    assert _view(d + "/mine_s.bam") == _view(d + "/ref_s.bam")
    # ... also through the spill-and-merge path and from a pipe
    monkeypatch.setenv("SSG_SORT_CHUNK_BYTES", "200000")
    monkeypatch.setenv("SSG_SORT_RANGES", "23")              # the merge goes over the genome in several stretches (a third of the budget each)
    with open(d + "/mine_u.bam", "rb") as fi:
        subprocess.run(sambamba + ["sort", "-t", "2", "-m", "1G", "--tmpdir=" + d + "/tmp2", "-o", d + "/mine_s2.bam", "/dev/stdin"], stdin=fi, check=True)
    monkeypatch.delenv("SSG_SORT_CHUNK_BYTES"); monkeypatch.delenv("SSG_SORT_RANGES")
    assert _view(d + "/mine_s2.bam") == _view(d + "/ref_s.bam")
    assert os.listdir(d + "/tmp2") == []
    # the sort left the index of its output and a note next to it: `index` recognises the pair as current (and drops the note); the index
    # computed from the file alone is the same bytes; a BAM that changed since is indexed again
    assert os.path.exists(d + "/mine_s.bam.bai") and os.path.exists(d + "/mine_s.bam.bai.ssg")
    by_sort = open(d + "/mine_s.bam.bai", "rb").read()
    subprocess.run(sambamba + ["index", d + "/mine_s.bam"], check=True)
    assert not os.path.exists(d + "/mine_s.bam.bai.ssg") and open(d + "/mine_s.bam.bai", "rb").read() == by_sort
    subprocess.run(sambamba + ["index", d + "/mine_s.bam"], check=True)                      # no note any more: computed from the file
    assert open(d + "/mine_s.bam.bai", "rb").read() == by_sort
    # the spill-and-merge path writes its index too (stretch by stretch, behind the merge's writer): what samtools computes from that file
    assert os.path.exists(d + "/mine_s2.bam.bai") and os.path.exists(d + "/mine_s2.bam.bai.ssg")
    os.rename(d + "/mine_s2.bam.bai", d + "/mine2.bai")
    subprocess.run([SAMTOOLS, "index", d + "/mine_s2.bam"], check=True)
    assert _parse_bai(d + "/mine2.bai") == _parse_bai(d + "/mine_s2.bam.bai")
    subprocess.run(sambamba + ["index", d + "/mine_s2.bam"], check=True)                     # note is stale now (the .bai was rewritten): computed from the file
    assert open(d + "/mine_s2.bam.bai", "rb").read() == open(d + "/mine2.bai", "rb").read()
    assert _view(d + "/mine_s2.bam", "20_slice:150000-151000") == _view(d + "/mine_s.bam", "20_slice:150000-151000")
    os.rename(d + "/mine_s.bam.bai", d + "/mine.bai")
    subprocess.run([SAMTOOLS, "index", d + "/mine_s.bam"], check=True)
    assert _parse_bai(d + "/mine.bai") == _parse_bai(d + "/mine_s.bam.bai")
    os.replace(d + "/mine.bai", d + "/mine_s.bam.bai")
    subprocess.run([SAMTOOLS, "index", d + "/ref_s.bam"], check=True)
    for region in ("20_slice:1000-5000", "20_slice:150000-151000", "20_slice:300000"):
        assert _view(d + "/mine_s.bam", region) == _view(d + "/ref_s.bam", region)
        assert len(_view(d + "/mine_s.bam", region).split("\n")) > 10
    # flagstat: the reference's samtools counts the same records; the sorted file is in order, the unsorted one is not
    mine = subprocess.check_output(sambamba + ["flagstat", "-t", "2", d + "/mine_s.bam"], text=True).split("\n")
    ref = subprocess.check_output([SAMTOOLS, "flagstat", d + "/ref_s.bam"], text=True)
    for key in ("in total", "secondary", "supplementary", "duplicates", "paired in sequencing", "read1", "read2", "properly paired", "with itself and mate mapped", "singletons"):
        a = [l for l in mine if key in l][0].split(" ")[0]; b = [l for l in ref.split("\n") if key in l][0].split(" ")[0]
        assert a == b, (key, a, b)
    assert [l for l in mine if "mapped" in l and "mate" not in l][0].split(" ")[0] == [l for l in ref.split("\n") if " mapped (" in l][0].split(" ")[0]
    assert [l for l in mine if "descents" in l][0].startswith("0 ")
    assert not [l for l in subprocess.check_output(sambamba + ["flagstat", d + "/mine_u.bam"], text=True).split("\n") if "descents" in l][0].startswith("0 ")
    # merge of two coordinate-sorted files
    (tmp_path / "second").mkdir()
    sam2 = _sam(tmp_path / "second", 300, seed=32, rg="h")      # another library: its own read group, as in `speedseq realign`
    subprocess.run([SAMTOOLS, "view", "-b", "-o", d + "/b2u.bam", sam2], check=True)
    subprocess.run([SAMTOOLS, "sort", "-o", d + "/b2.bam", d + "/b2u.bam"], check=True)
    subprocess.run(sambamba + ["merge", "-t", "2", d + "/mine_m.bam", d + "/ref_s.bam", d + "/b2.bam"], check=True)
    subprocess.run([SAMTOOLS, "merge", "-f", d + "/ref_m.bam", d + "/ref_s.bam", d + "/b2.bam"], check=True)
    def body(t):     # samtools merge re-appends the RG tag it translates: compare with the optional fields in canonical order
        out = []
        for l in t.split("\n"):
            if l and not l.startswith("@"):
                f = l.split("\t")
                out.append(f[:11] + sorted(f[11:]))
        return out
    assert body(_view(d + "/mine_m.bam")) == body(_view(d + "/ref_m.bam"))
    hdr = _view(d + "/mine_m.bam")
    assert "@RG\tID:g\t" in hdr and "@RG\tID:h\t" in hdr and "SO:coordinate" in hdr


def test_sambamba_emu_matches_samtools(tmp_path, emu_lib, monkeypatch):
    _check([os.path.join(ROOT, "tests", "emu", "sambamba_emu")], tmp_path, monkeypatch)


def test_sambamba_emu_device_deflate_matches_samtools(tmp_path, emu_lib, monkeypatch):
    """the sorted file's BGZF blocks deflated by the device kernel (k_bgzf.h; the emulation only does so on request): the reference's
    samtools must read the file -- records, order, index, region queries -- exactly as it reads the zlib-written one"""
    monkeypatch.setenv("SSG_BGZF_DEVICE", "1")
    _check([os.path.join(ROOT, "tests", "emu", "sambamba_emu")], tmp_path, monkeypatch)


def test_sambamba_emu_sort_to_a_pipe(tmp_path, emu_lib, monkeypatch):
    """`-o /dev/stdout` into a pipe: no index can be written (no offsets), the records are the same, in memory and through the spill-and-merge path"""
    d = str(tmp_path)
    sam = _sam(tmp_path, 600, seed=36)
    sambamba = os.path.join(ROOT, "tests", "emu", "sambamba_emu")
    with open(sam, "rb") as fi, open(d + "/u.bam", "wb") as fo:
        subprocess.run([sambamba, "view", "-S", "-f", "bam", "-l", "0", "/dev/stdin"], stdin=fi, stdout=fo, check=True)
    subprocess.run([sambamba, "sort", "-t", "3", "-m", "1G", "--tmpdir=" + d + "/t0", "-o", d + "/file.bam", d + "/u.bam"], check=True)
    for k, env in enumerate(({}, {"SSG_SORT_CHUNK_BYTES": "150000", "SSG_SORT_RANGES": "7"})):
        out = "%s/piped%d.bam" % (d, k)
        subprocess.run("%s sort -t 3 -m 1G --tmpdir=%s/t%d -o /dev/stdout %s/u.bam | cat > %s" % (sambamba, d, k + 1, d, out), shell=True, check=True, env=dict(os.environ, **env))
        assert _view(out) == _view(d + "/file.bam")
        assert os.listdir("%s/t%d" % (d, k + 1)) == []


def test_sambamba_emu_device_deflate_on_several_devices(tmp_path, emu_lib, monkeypatch):
    """every visible device deflates blocks of the sorted file (producer t on device t mod N, the writer and the index thread follow whoever made a
    block): three emulated devices, batches of 16 blocks so that a small file keeps several producers busy; same checks as above"""
    monkeypatch.setenv("SSG_BGZF_DEVICE", "1")
    monkeypatch.setenv("SSG_EMU_DEVICES", "3")
    monkeypatch.setenv("SSG_SORT_DEV_BATCH", "16")
    monkeypatch.setenv("SSG_DEBUG", "1")
    d = str(tmp_path)
    sam = _sam(tmp_path, 1500, seed=35)
    sambamba = os.path.join(ROOT, "tests", "emu", "sambamba_emu")
    with open(sam, "rb") as fi, open(d + "/u.bam", "wb") as fo:
        subprocess.run([sambamba, "view", "-S", "-f", "bam", "-l", "0", "/dev/stdin"], stdin=fi, stdout=fo, check=True)
    r = subprocess.run([sambamba, "sort", "-t", "4", "-m", "1G", "--tmpdir=" + d + "/tmp", "-o", d + "/s.bam", d + "/u.bam"], check=True, capture_output=True, text=True)
    assert "blocks deflated on 3 device(s)" in r.stderr, r.stderr[-500:]
    subprocess.run([SAMTOOLS, "view", "-b", "-u", "-o", d + "/ref_u.bam", sam], check=True)
    subprocess.run([SAMTOOLS, "sort", "-o", d + "/ref_s.bam", d + "/ref_u.bam"], check=True)
    assert _view(d + "/s.bam") == _view(d + "/ref_s.bam")
    os.rename(d + "/s.bam.bai", d + "/mine.bai")
    subprocess.run([SAMTOOLS, "index", d + "/s.bam"], check=True)
    assert _parse_bai(d + "/mine.bai") == _parse_bai(d + "/s.bam.bai")


@pytest.mark.parametrize("fail_after", ["0", "3"])
def test_sambamba_emu_device_deflate_failure_falls_back_to_the_host(tmp_path, emu_lib, monkeypatch, fail_after):
    """a device that fails while the sorted file is being written (before its first batch, or after two) does not end the sort: the producers stop and the
    host's pool compresses the groups that are not there yet; samtools reads the same records in the same order, and the index the sort wrote is the file's"""
    monkeypatch.setenv("SSG_BGZF_DEVICE", "1")
    monkeypatch.setenv("SSG_EMU_DEVICES", "2")
    monkeypatch.setenv("SSG_SORT_DEV_BATCH", "16")
    monkeypatch.setenv("SSG_BGZF_FAIL_AFTER", fail_after)
    d = str(tmp_path)
    sam = _sam(tmp_path, 20000, seed=37)   # ~100 blocks: seven batches of 16
    sambamba = os.path.join(ROOT, "tests", "emu", "sambamba_emu")
    with open(sam, "rb") as fi, open(d + "/u.bam", "wb") as fo:
        subprocess.run([sambamba, "view", "-S", "-f", "bam", "-l", "0", "/dev/stdin"], stdin=fi, stdout=fo, check=True)
    r = subprocess.run([sambamba, "sort", "-t", "4", "-m", "1G", "--tmpdir=" + d + "/tmp", "-o", d + "/s.bam", d + "/u.bam"], check=True, capture_output=True, text=True, timeout=300)
    assert "compressing the rest of the output on the host" in r.stderr, r.stderr[-500:]
    subprocess.run([SAMTOOLS, "view", "-b", "-u", "-o", d + "/ref_u.bam", sam], check=True)
    subprocess.run([SAMTOOLS, "sort", "-o", d + "/ref_s.bam", d + "/ref_u.bam"], check=True)
    assert _view(d + "/s.bam") == _view(d + "/ref_s.bam")
    os.rename(d + "/s.bam.bai", d + "/mine.bai")
    subprocess.run([SAMTOOLS, "index", d + "/s.bam"], check=True)
    assert _parse_bai(d + "/mine.bai") == _parse_bai(d + "/s.bam.bai")


@pytest.mark.gpu
@pytest.mark.parametrize("device_deflate", ["1", "0"])
def test_sambamba_gpu_matches_samtools(tmp_path, gpu_lib, monkeypatch, device_deflate):
    monkeypatch.setenv("SSG_BGZF_DEVICE", device_deflate)   # 1: the default on a GPU (k_bgzf.h); 0: zlib on the host's threads
    _check([os.path.join(ROOT, "bin", "sambamba")], tmp_path, monkeypatch)


def test_sambamba_emu_stream_of_many_blocks(tmp_path, emu_lib):
    """`view -S -f bam -l 0 | sort` as the reference wires it (bin/speedseq:440-441) on a stream of > 512 BGZF blocks of 64 KB
    (45 MB): the reader's read-ahead buffer is compacted between batches of blocks, not inside one (it used to be: 'inflate failed'
    on any BAM with more than ~32 MB per 512 blocks, i.e. every real one at compression level 0)"""
    sambamba = os.path.join(ROOT, "tests", "emu", "sambamba_emu")
    sam = _sam(tmp_path, n_pairs=1500, seed=33)
    lines = open(sam).read().split("\n")
    hdr = [l for l in lines if l.startswith("@")]
    body = [l for l in lines if l and not l.startswith("@")]
    big = str(tmp_path / "big.sam")
    with open(big, "w") as f:
        f.write("\n".join(hdr) + "\n")
        for k in range(45):
            f.write("\n".join(body) + "\n")
    out = str(tmp_path / "sorted.bam")
    subprocess.check_call("%s view -S -f bam -l 0 %s | %s sort -t 4 -m 8G --tmpdir=%s -o %s /dev/stdin" % (sambamba, big, sambamba, str(tmp_path), out), shell=True)
    subprocess.check_call([sambamba, "index", out])
    assert os.path.getsize(big) > 40 << 20
    assert int(subprocess.check_output([SAMTOOLS, "view", "-c", out])) == 45 * len(body)
    ref = str(tmp_path / "ref.bam")
    subprocess.check_call("%s view -b -u %s | %s sort -o %s -" % (SAMTOOLS, big, SAMTOOLS, ref), shell=True, stderr=subprocess.DEVNULL)
    assert _view(out).split("\n")[len(hdr):] == _view(ref).split("\n")[len(hdr):] or \
        [l for l in _view(out).split("\n") if not l.startswith("@")] == [l for l in _view(ref).split("\n") if not l.startswith("@")]


@pytest.mark.parametrize("host_batches,fail_after", [("1", ""), ("2", ""), ("1", "2")])
def test_sambamba_emu_device_and_host_share_the_blocks(tmp_path, emu_lib, monkeypatch, host_batches, fail_after):
    """batch k of the sorted file belongs to slot k mod (device producers + host slots): the device's producers and the host's zlib pool write one file, the
    same bytes on every run (the rule is fixed, not first come first served); with a device that fails on the way the host finishes alone"""
    monkeypatch.setenv("SSG_BGZF_DEVICE", "1")
    monkeypatch.setenv("SSG_EMU_DEVICES", "2")
    monkeypatch.setenv("SSG_SORT_DEV_BATCH", "16")
    monkeypatch.setenv("SSG_SORT_HOST_BATCHES", host_batches)
    monkeypatch.setenv("SSG_DEBUG", "1")
    if fail_after:
        monkeypatch.setenv("SSG_BGZF_FAIL_AFTER", fail_after)
    d = str(tmp_path)
    sam = _sam(tmp_path, 20000, seed=39)   # ~100 blocks: seven batches of 16
    sambamba = os.path.join(ROOT, "tests", "emu", "sambamba_emu")
    with open(sam, "rb") as fi, open(d + "/u.bam", "wb") as fo:
        subprocess.run([sambamba, "view", "-S", "-f", "bam", "-l", "0", "/dev/stdin"], stdin=fi, stdout=fo, check=True)
    outs = []
    for k in range(1 if fail_after else 2):
        r = subprocess.run([sambamba, "sort", "-t", "4", "-m", "1G", "--tmpdir=" + d + "/tmp", "-o", "%s/s%d.bam" % (d, k), d + "/u.bam"], check=True, capture_output=True, text=True, timeout=300)
        if fail_after:
            assert "compressing the rest of the output on the host" in r.stderr, r.stderr[-500:]
        else:
            assert "batches by the host's pool" in r.stderr, r.stderr[-800:]
        outs.append(open("%s/s%d.bam" % (d, k), "rb").read())
    assert len(set(outs)) == 1
    subprocess.run([SAMTOOLS, "view", "-b", "-u", "-o", d + "/ref_u.bam", sam], check=True)
    subprocess.run([SAMTOOLS, "sort", "-o", d + "/ref_s.bam", d + "/ref_u.bam"], check=True)
    assert _view(d + "/s0.bam") == _view(d + "/ref_s.bam")
    os.rename(d + "/s0.bam.bai", d + "/mine.bai")
    subprocess.run([SAMTOOLS, "index", d + "/s0.bam"], check=True)
    assert _parse_bai(d + "/mine.bai") == _parse_bai(d + "/s0.bam.bai")
