"""BASELINE.json configs[0]: the reference's own `speedseq align` script (bin/speedseq:189-504), run UNMODIFIED with a
speedseq.config that names this repo's executables (the reference's plugin mechanism, bin/speedseq.config:13-14):
  * the CPU oracle's `bwa` / `samblaster`        (plumbing baseline),
  * the host-emulation build of the product's    (same sources as bin/bwa, bin/samblaster; CPU box),
  * the product executables on the MI355X        (-m gpu),
the product runs with this repo's native sambamba (SAM->BAM, device coordinate sort, BGZF, BAI), the oracle run with the reference's
samtools behind sambamba's command line (tools/sambamba_samtools_shim.sh) as the comparator; bin/parallel throughout.  The three BAMs of the product runs must decode (samtools view) to the
same records as the oracle run's.  The script comes from /root/reference when present, else from the fixture copy
tests/golden/speedseq_ref_script.sh (tests/golden/make_golden.sh)."""
import os
import shutil
import subprocess

import pytest

import simreads
from common import EXAMPLE_FA, ROOT

REF_SCRIPT = "/root/reference/bin/speedseq" if os.path.exists("/root/reference/bin/speedseq") else os.path.join(ROOT, "tests", "golden", "speedseq_ref_script.sh")
SAMTOOLS = os.path.join(ROOT, "oracle", "_ref", "samtools")
ORC = os.path.join(ROOT, "oracle", "orc_bwa")
EMU = os.path.join(ROOT, "tests", "emu")
SHIM = os.path.join(ROOT, "tools", "sambamba_samtools_shim.sh")   # comparator: the reference's samtools behind sambamba's command line


def _need_tools():
    if not os.path.exists(SAMTOOLS) and os.path.exists("/root/reference"):
        subprocess.check_call(["sh", os.path.join(ROOT, "oracle", "build_ref_tools.sh")])
    if not os.path.exists(SAMTOOLS) or shutil.which("mawk") is None:
        pytest.skip("samtools (oracle/_ref) or mawk unavailable")
    if not os.path.exists(ORC):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


def _run_align(d, bwa_cmd, samblaster_cmd, fq, n_threads=4, sambamba=SHIM, fq2=None, rg="@RG\\tID:NA12878\\tSM:NA12878\\tLB:lib1", config_extra="", env_extra=None):
    """runs the reference script in directory d with wrappers around the given executables; returns the output prefix"""
    os.makedirs(d)
    bindir = os.path.join(d, "bin")
    os.makedirs(bindir)
    for name, cmd in (("bwa", bwa_cmd), ("samblaster", samblaster_cmd)):
        with open(os.path.join(bindir, name), "w") as f:
            f.write("#!/bin/sh\nexec %s \"$@\"\n" % cmd)
        os.chmod(os.path.join(bindir, name), 0o755)
    os.symlink(shutil.which("mawk"), os.path.join(bindir, "gawk"))        # the script hard-codes `gawk`
    cfg = os.path.join(d, "speedseq.config")
    with open(cfg, "w") as f:
        f.write("BWA=%s/bwa\nSAMBLASTER=%s/samblaster\nSAMBAMBA=%s\nPARALLEL=%s/bin/parallel\n%s" % (bindir, bindir, sambamba, ROOT, config_extra))
    ref = os.path.join(d, "ref.fa")
    shutil.copy(EXAMPLE_FA, ref)   # no index next to it: the script must call `$BWA index`
    env = dict(os.environ, PATH="%s:%s" % (bindir, os.environ["PATH"]))
    env.update(env_extra or {})
    out = os.path.join(d, "example")
    r = subprocess.run(["bash", REF_SCRIPT, "align", "-K", cfg, "-o", out, "-M", "3", "-t", str(n_threads)] + ([] if fq2 else ["-p"]) +
                       ["-R", rg, ref, fq] + ([fq2] if fq2 else []),
                       cwd=d, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for ext in ("amb", "ann", "bwt", "pac", "sa"):     # `$BWA index` ran and wrote upstream's bytes
        assert open(ref + "." + ext, "rb").read() == open(EXAMPLE_FA + "." + ext, "rb").read(), ext
    return out


def _view(bam):
    txt = subprocess.check_output([SAMTOOLS, "view", "-h", bam], text=True)
    return [l for l in txt.split("\n") if not l.startswith("@PG")]


def _check_outputs(out):
    for suffix in (".bam", ".splitters.bam", ".discordants.bam"):
        assert os.path.getsize(out + suffix) > 0
        assert os.path.exists(out + suffix + ".bai")
    assert int(subprocess.check_output([SAMTOOLS, "view", "-c", out + ".bam"])) >= 3000
    assert int(subprocess.check_output([SAMTOOLS, "view", "-c", "-f", "1024", out + ".bam"])) > 50     # simulated 5 % duplicate fragments
    hdr = subprocess.check_output([SAMTOOLS, "view", "-H", out + ".bam"], text=True)
    assert "SO:coordinate" in hdr and "@RG\tID:NA12878" in hdr
    assert int(subprocess.check_output([SAMTOOLS, "view", "-c", out + ".splitters.bam"])) > 0
    assert int(subprocess.check_output([SAMTOOLS, "view", "-c", out + ".discordants.bam"])) > 0
    rec = subprocess.check_output([SAMTOOLS, "view", out + ".splitters.bam"], text=True).split("\n")[0].split("\t")
    assert rec[9] == "*" and rec[10] == "*" and (rec[0].endswith("_1") or rec[0].endswith("_2"))


def _fastq(tmp_path, n_pairs=1500):
    fq = str(tmp_path / "reads.fq.gz")
    simreads.write_fastq(fq, simreads.simulate(simreads.read_fasta(EXAMPLE_FA), n_pairs, seed=11))
    return fq


def _compare(out, exp):
    for suffix in (".bam", ".splitters.bam", ".discordants.bam"):
        assert _view(out + suffix) == _view(exp + suffix), suffix


def test_reference_align_script_with_oracle_tools(tmp_path):
    _need_tools()
    out = _run_align(str(tmp_path / "orc"), ORC, ORC + " samblaster", _fastq(tmp_path))
    _check_outputs(out)


def test_reference_align_script_with_product_sources_emulated(tmp_path, emu_lib):
    """the product's bwa / samblaster host code + kernel sources (host-emulation build) behind the unmodified script"""
    _need_tools()
    fq = _fastq(tmp_path)
    exp = _run_align(str(tmp_path / "orc"), ORC, ORC + " samblaster", fq)
    out = _run_align(str(tmp_path / "emu"), os.path.join(EMU, "bwa_emu"), os.path.join(EMU, "samblaster_emu"), fq, sambamba=os.path.join(EMU, "sambamba_emu"))
    _check_outputs(out)
    _compare(out, exp)


@pytest.mark.gpu
def test_reference_align_script_with_product_executables(tmp_path, gpu_lib):
    """the product executables on the MI355X behind the unmodified script; BAMs decode-equal to the oracle run's"""
    _need_tools()
    fq = _fastq(tmp_path, 4000)
    exp = _run_align(str(tmp_path / "orc"), ORC, ORC + " samblaster", fq)
    out = _run_align(str(tmp_path / "gpu"), os.path.join(ROOT, "bin", "bwa"), os.path.join(ROOT, "bin", "samblaster"), fq, sambamba=os.path.join(ROOT, "bin", "sambamba"))
    _check_outputs(out)
    _compare(out, exp)


def test_reference_align_script_two_fastq_files_emulated(tmp_path, emu_lib):
    """`speedseq align ref.fa in1.fq.gz in2.fq.gz` (bin/speedseq:468-469: `bwa mem` without -p on two files)"""
    _need_tools()
    import gzip
    pairs = simreads.simulate(simreads.read_fasta(EXAMPLE_FA), 700, seed=12)
    f1, f2 = str(tmp_path / "r1.fq.gz"), str(tmp_path / "r2.fq.gz")
    for path, which in ((f1, 1), (f2, 2)):
        with gzip.open(path, "wt") as f:
            for name, r1, r2 in pairs:
                r = r1 if which == 1 else r2
                f.write("@%s/%d\n%s\n+\n%s\n" % (name, which, "".join("ACGTN"[c] for c in r), "I" * len(r)))
    exp = _run_align(str(tmp_path / "orc"), ORC, ORC + " samblaster", f1, fq2=f2)
    out = _run_align(str(tmp_path / "emu"), os.path.join(EMU, "bwa_emu"), os.path.join(EMU, "samblaster_emu"), f1, sambamba=os.path.join(EMU, "sambamba_emu"), fq2=f2)
    _check_outputs_min(out)
    _compare(out, exp)


def _check_outputs_min(out):
    for suffix in (".bam", ".splitters.bam", ".discordants.bam"):
        assert os.path.getsize(out + suffix) > 0 and os.path.exists(out + suffix + ".bai")
    assert int(subprocess.check_output([SAMTOOLS, "view", "-c", out + ".bam"])) >= 1400


# ---- BASELINE.json configs[0] with the reference's own read simulator ----
WGSIM = os.path.join(ROOT, "oracle", "_ref", "wgsim")


def _wgsim_fastq(tmp_path, n_pairs, seed=11):
    """example/run_speedseq.sh:4-10 aligns data/NA12878.20slice.30X.fastq.gz, which the reference tree lists as a missing blob; SURVEY.md 8d prescribes its
    stand-in: 30x of the example FASTA from the reference's wgsim (src/samtools-1.3.1/misc/wgsim.c:438-463), interleaved and gzipped"""
    import gzip
    if not os.path.exists(WGSIM) and os.path.exists("/root/reference"):
        subprocess.check_call(["sh", os.path.join(ROOT, "oracle", "build_ref_tools.sh")])
    if not os.path.exists(WGSIM):
        pytest.skip("the reference's wgsim (oracle/_ref) is not built")
    r1, r2 = str(tmp_path / "w1.fq"), str(tmp_path / "w2.fq")
    subprocess.check_call([WGSIM, "-S", str(seed), "-N", str(n_pairs), "-1", "150", "-2", "150", "-d", "400", "-s", "50", "-e", "0.005", "-r", "0.001", "-R", "0.15", "-X", "0.3",
                           EXAMPLE_FA, r1, r2], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    fq = str(tmp_path / "wgsim.fq.gz")
    with open(r1) as a, open(r2) as b, gzip.open(fq, "wt", compresslevel=1) as out:
        while True:
            x = [a.readline() for _ in range(4)]
            y = [b.readline() for _ in range(4)]
            if not x[0] or not y[0]:
                break
            out.write("".join(x)); out.write("".join(y))
    return fq


def _check_wgsim_outputs(out, n_pairs):
    for suffix in (".bam", ".splitters.bam", ".discordants.bam"):
        assert os.path.getsize(out + suffix) > 0 and os.path.exists(out + suffix + ".bai")
    assert int(subprocess.check_output([SAMTOOLS, "view", "-c", "-F", "2304", out + ".bam"])) == 2 * n_pairs      # every read once as a primary line
    assert "SO:coordinate" in subprocess.check_output([SAMTOOLS, "view", "-H", out + ".bam"], text=True)


def test_reference_align_script_on_wgsim_reads_emulated(tmp_path, emu_lib):
    """config 1 at a size the host emulation finishes in a minute: reads of the reference's wgsim, the unmodified script, product sources vs the oracle's executables, three BAMs"""
    _need_tools()
    fq = _wgsim_fastq(tmp_path, 1500)
    exp = _run_align(str(tmp_path / "orc"), ORC, ORC + " samblaster", fq)
    out = _run_align(str(tmp_path / "emu"), os.path.join(EMU, "bwa_emu"), os.path.join(EMU, "samblaster_emu"), fq, sambamba=os.path.join(EMU, "sambamba_emu"))
    _check_wgsim_outputs(out, 1500)
    _compare(out, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_reference_align_script_on_wgsim_reads_config1(tmp_path, gpu_lib, fused):
    """config 1 as SURVEY.md 8d states it -- wgsim -S 11 -N 32164 (30x of the 321 635-base example) -- on the MI355X, text and fused hand-off, against the oracle's run"""
    _need_tools()
    fq = _wgsim_fastq(tmp_path, 32164)
    exp = _run_align(str(tmp_path / "orc"), ORC, ORC + " samblaster", fq, n_threads=8)
    out = _run_align(str(tmp_path / "gpu"), os.path.join(ROOT, "bin", "bwa"), os.path.join(ROOT, "bin", "samblaster"), fq, sambamba=os.path.join(ROOT, "bin", "sambamba"),
                     config_extra="export SSG_FUSED=1\n" if fused else "")
    _check_wgsim_outputs(out, 32164)
    _compare(out, exp)
