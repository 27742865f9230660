"""BASELINE.json configs[0]: the reference's own `speedseq align` script (bin/speedseq:189-504), run
UNMODIFIED from /root/reference with a speedseq.config that names this repo's executables -- here
the CPU oracle builds of `bwa` / `samblaster` plus the sambamba / parallel shims (CPU plumbing, no
GPU).  Skipped where the reference checkout is absent (e.g. on the GPU box)."""
import gzip
import os
import shutil
import subprocess

import pytest

import simreads
from common import EXAMPLE_FA, ROOT

REF_SCRIPT = "/root/reference/bin/speedseq"
SAMTOOLS = os.path.join(ROOT, "oracle", "_ref", "samtools")


@pytest.mark.skipif(not os.path.exists(REF_SCRIPT), reason="reference checkout not present")
def test_reference_align_script_with_oracle_tools(tmp_path):
    if not os.path.exists(SAMTOOLS):
        subprocess.check_call(["sh", os.path.join(ROOT, "oracle", "build_ref_tools.sh")])
    if not os.path.exists(SAMTOOLS) or shutil.which("mawk") is None:
        pytest.skip("samtools (oracle/_ref) or mawk unavailable")
    orc = os.path.join(ROOT, "oracle", "orc_bwa")
    if not os.path.exists(orc):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    bindir = tmp_path / "bin"
    bindir.mkdir()
    (bindir / "bwa").write_text("#!/bin/sh\nexec %s \"$@\"\n" % orc)
    (bindir / "samblaster").write_text("#!/bin/sh\nexec %s samblaster \"$@\"\n" % orc)
    for f in ("bwa", "samblaster"):
        os.chmod(bindir / f, 0o755)
    os.symlink(shutil.which("mawk"), bindir / "gawk")        # the script hard-codes `gawk`
    cfg = tmp_path / "speedseq.config"
    cfg.write_text("BWA=%s/bwa\nSAMBLASTER=%s/samblaster\nSAMBAMBA=%s/bin/sambamba\nPARALLEL=%s/bin/parallel\n" % (bindir, bindir, ROOT, ROOT))
    ref = tmp_path / "ref.fa"
    shutil.copy(EXAMPLE_FA, ref)   # no index next to it: the script must call `$BWA index`
    contigs = simreads.read_fasta(str(ref))
    pairs = simreads.simulate(contigs, 1500, seed=11)
    fq = tmp_path / "reads.fq.gz"
    simreads.write_fastq(str(fq), pairs)
    env = dict(os.environ, PATH="%s:%s" % (bindir, os.environ["PATH"]))
    out = tmp_path / "example"
    r = subprocess.run(["bash", REF_SCRIPT, "align", "-K", str(cfg), "-o", str(out), "-M", "3", "-t", "4", "-p",
                        "-R", "@RG\\tID:NA12878\\tSM:NA12878\\tLB:lib1", str(ref), str(fq)],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for suffix in (".bam", ".splitters.bam", ".discordants.bam"):
        assert os.path.getsize(str(out) + suffix) > 0
        assert os.path.exists(str(out) + suffix + ".bai")
    n = int(subprocess.check_output([SAMTOOLS, "view", "-c", str(out) + ".bam"]))
    assert n >= 3000
    dups = int(subprocess.check_output([SAMTOOLS, "view", "-c", "-f", "1024", str(out) + ".bam"]))
    assert dups > 50                                            # simulated 5 % duplicate fragments
    hdr = subprocess.check_output([SAMTOOLS, "view", "-H", str(out) + ".bam"], text=True)
    assert "SO:coordinate" in hdr and "@RG\tID:NA12878" in hdr
    assert int(subprocess.check_output([SAMTOOLS, "view", "-c", str(out) + ".splitters.bam"])) > 0
    assert int(subprocess.check_output([SAMTOOLS, "view", "-c", str(out) + ".discordants.bam"])) > 0
    rec = subprocess.check_output([SAMTOOLS, "view", str(out) + ".splitters.bam"], text=True).split("\n")[0].split("\t")
    assert rec[9] == "*" and rec[10] == "*" and (rec[0].endswith("_1") or rec[0].endswith("_2"))
