"""`bwa index` on the device (speedseq_amd/csrc/k_index.h + ssg_index_build.cpp) reproduces the reference's bundled
index bytes (tests/golden/chr20_slice.fa.* = /root/reference/example/data/*.fasta.*) and, on references with
ambiguity holes / repeats / homopolymer ends, the oracle's restatement of upstream `bwa index`."""
import filecmp
import os
import shutil

import numpy as np
import pytest

from common import EXAMPLE_FA

EXTS = ("bwt", "sa", "pac", "ann", "amb")


def tricky_fasta(path, seed):
    rng = np.random.default_rng(seed)

    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    rep = rnd(300)
    ctgs = [("c1 first contig comment", "T" * 42 + rnd(500) + rep + "N" * 10 + rnd(100) + rep + "NRYNN" + rnd(77) + "AC" * 200 + rep.lower()),
            ("c2", "A" * 150 + rnd(333) + "N" + "T" * 100),
            ("c3\tx y", rnd(1000) + rep[:150] + rnd(10) + "A" * 76)]
    with open(path, "w") as f:
        for n, s in ctgs:
            f.write(">" + n + "\n")
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")


def _golden(lib, tmp_path):
    fa = str(tmp_path / "x.fa")
    shutil.copy(EXAMPLE_FA, fa)
    h = lib.index_build_fasta(fa)
    lib.index_save(h, fa)
    lib.index_destroy(h)
    for ext in EXTS:
        assert filecmp.cmp(fa + "." + ext, EXAMPLE_FA + "." + ext, shallow=False), ext


def _vs_oracle(lib, oracle, tmp_path, monkeypatch):
    for seed in (1, 2):
        fa, ofa = str(tmp_path / ("r%d.fa" % seed)), str(tmp_path / ("o%d.fa" % seed))
        tricky_fasta(fa, seed)
        shutil.copy(fa, ofa)
        oracle.idx_build(ofa, save=True)
        for p in ("0", "1", "2"):           # bucket prefix lengths: one bucket, 4, 16
            monkeypatch.setenv("SSG_INDEX_BUCKET_P", p)
            h = lib.index_build_fasta(fa)
            lib.index_save(h, fa)
            lib.index_destroy(h)
            for ext in EXTS:
                assert filecmp.cmp(fa + "." + ext, ofa + "." + ext, shallow=False), (seed, p, ext)


def test_index_build_emu_matches_golden(emu_lib, tmp_path, monkeypatch):
    monkeypatch.setenv("SSG_INDEX_MAX_WG", "7")     # grid-stride path of every construction kernel (read once per process)
    _golden(emu_lib, tmp_path)


def test_index_build_emu_bucketed_golden(emu_lib, tmp_path, monkeypatch):
    monkeypatch.setenv("SSG_INDEX_BUCKET_P", "3")    # 64 buckets: a bucket's suffixes listed by ssg_k_idx_bucket_count / _scatter (k_index.h)
    _golden(emu_lib, tmp_path)


def test_index_build_emu_matches_oracle_on_holes_and_repeats(emu_lib, oracle, tmp_path, monkeypatch):
    _vs_oracle(emu_lib, oracle, tmp_path, monkeypatch)


@pytest.mark.gpu
def test_index_build_gpu_matches_golden(gpu_lib, tmp_path):
    _golden(gpu_lib, tmp_path)


@pytest.mark.gpu
def test_index_build_gpu_matches_oracle_on_holes_and_repeats(gpu_lib, oracle, tmp_path, monkeypatch):
    _vs_oracle(gpu_lib, oracle, tmp_path, monkeypatch)


@pytest.mark.gpu
def test_index_build_gpu_bucketed_golden(gpu_lib, tmp_path, monkeypatch):
    monkeypatch.setenv("SSG_INDEX_BUCKET_P", "3")
    _golden(gpu_lib, tmp_path)


@pytest.mark.gpu
def test_bwa_index_executable(tmp_path):
    """the `bwa index` the reference runs at bin/speedseq:389 (product executable, native: no Python behind it)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fa = str(tmp_path / "x.fa")
    shutil.copy(EXAMPLE_FA, fa)
    subprocess.check_call([os.path.join(root, "bin", "bwa"), "index", fa])
    for ext in EXTS:
        assert filecmp.cmp(fa + "." + ext, EXAMPLE_FA + "." + ext, shallow=False), ext
