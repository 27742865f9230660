"""Oracle pinning: the oracle's `bwa index` restatement reproduces the reference's bundled index
byte for byte (tests/golden/chr20_slice.fa.* = /root/reference/example/data/*.fasta.*)."""
import filecmp
import os
import shutil

from common import EXAMPLE_FA


def test_oracle_index_matches_golden(oracle, tmp_path):
    fa = str(tmp_path / "ref.fa")
    shutil.copy(EXAMPLE_FA, fa)
    oracle.idx_build(fa, save=True)
    for ext in ("amb", "ann", "bwt", "pac", "sa"):
        assert filecmp.cmp(fa + "." + ext, EXAMPLE_FA + "." + ext, shallow=False), ext


def test_oracle_loads_golden_index(oracle):
    idx = oracle.idx_load(EXAMPLE_FA)
    assert idx
