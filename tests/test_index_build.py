"""The GPU-side FM-index builder (torch plumbing) reproduces the reference's bundled index bytes."""
import filecmp

import numpy as np
import torch

import simreads
from common import EXAMPLE_FA
from speedseq_amd import index_build


def _build(tmp_path, dev):
    contigs = simreads.read_fasta(EXAMPLE_FA)
    fwd = torch.from_numpy(np.concatenate([s for _, s in contigs])).to(dev)
    ix = index_build.build_index_arrays(fwd)
    prefix = str(tmp_path / "x.fa")
    index_build.write_index_files(prefix, ix, [n for n, _ in contigs], [len(s) for _, s in contigs])
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(prefix + "." + ext, EXAMPLE_FA + "." + ext, shallow=False), ext


def test_index_build_cpu_matches_golden(tmp_path):
    _build(tmp_path, "cpu")
