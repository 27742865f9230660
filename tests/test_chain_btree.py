"""The container of upstream mem_chain: klib's B-tree (kbtree.h, t = 5: nine chains a node), keyed by chain position with equal keys allowed.

Until round 6 oracle and kernels both kept a read's chains in position order with the semantics of ONE leaf of that tree -- exact for reads with at most nine
chains or without two chains at one position, a shared deviation otherwise that no test could see.  Now the oracle restates the tree itself (oracle/orc_mem.c; [RECALL]:
kbtree.h is not in the reference tree), every chaining kernel flags the reads that can differ, and the flagged reads are chained again on a device copy of the tree
(csrc/k_chain.h ssg_k_chain_kb).  Here: reads constructed to differ (a unit repeated in the read more than a band apart + many short hits elsewhere), counted by
the oracle's exposure function, and aligned by every chaining form of the device against the oracle's tree."""
import os

import numpy as np
import pytest

import common
import simreads


def _constructed_reads(prefix, n, seed, L=250):
    ref = simreads.read_fasta(prefix)[0][1]
    rng = np.random.default_rng(seed)
    reads = []
    for _ in range(n):
        u0 = int(rng.integers(0, len(ref) - 40))
        U = ref[u0:u0 + int(rng.integers(25, 35))]
        segs = [U]
        while sum(len(x) for x in segs) < L - len(U) - 5:
            p = int(rng.integers(0, len(ref) - 30))
            segs.append(ref[p:p + int(rng.integers(20, 28))])
        segs.append(U)
        reads.append(np.concatenate(segs)[:L].astype(np.uint8))
    return reads


@pytest.fixture(scope="module")
def repeat_ref(oracle, tmp_path_factory):
    d = str(tmp_path_factory.mktemp("kbref"))
    return common.repeat_reference(oracle, d)


def test_oracle_tree_and_array_differ_only_where_they_can(oracle, repeat_ref):
    oidx = oracle.idx_load(repeat_ref)
    reads = _constructed_reads(repeat_ref, 300, 3) + list(common.reads_with_inner_repeats(repeat_ref, 150, 9, rl=250))
    off = np.zeros(len(reads) + 1, dtype=np.int64); off[1:] = np.cumsum([len(r) for r in reads]); seq = np.concatenate(reads)
    d, which = oracle.chain_exposure(oidx, seq, off, per_read=True)
    assert d["differ"] > 50 and d["gt9_and_dup"] >= d["differ"], d             # the constructed reads do part the two containers ...
    assert not np.any((which & 1) & ~((which >> 1) & 1)), "a read without more than 9 chains and a shared position differs"   # ... and only flagged reads do
    # simulated reads of the bundled slice: a thousand reads with more than nine chains, none that differs
    pairs, seqs, seq2, off2 = common.sim_reads(2000, 5, 150)
    d2 = oracle.chain_exposure(oracle.idx_load(common.EXAMPLE_FA), seq2, off2)
    assert d2["gt9"] > 100 and d2["differ"] == 0, d2


def _check_forms(lib, oracle, prefix, reads):
    envs = [{}, {"SSG_CHAIN_LDS": "0"}, {"SSG_CHAIN_WAVE_MIN": "8"}, {"SSG_CHAIN_WAVE_MIN": "8", "SSG_CHAIN_RANKED": "0"}, {"SSG_CHAIN_WAVE_MIN": "100000"}]
    n_flagged = 0
    for env in envs:       # the light reads' LDS kernel / the global-memory lane kernel / the wave kernel's ranked and shifting forms / everything one lane each
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            n_flagged = max(n_flagged, common.check_align1_reads(lib, oracle, reads, prefix=prefix))
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
    return n_flagged


def test_emu_every_chaining_form_follows_the_tree(emu_lib, oracle, repeat_ref):
    reads = _constructed_reads(repeat_ref, 120, 4) + list(common.reads_with_inner_repeats(repeat_ref, 40, 10, rl=250))
    _check_forms(emu_lib, oracle, repeat_ref, reads)
    # without the second pass the kernels keep the array's order and the same comparison fails: the flag is what closes the gap
    os.environ["SSG_CHAIN_KBTREE"] = "0"
    try:
        with pytest.raises(AssertionError):
            common.check_align1_reads(emu_lib, oracle, reads, prefix=repeat_ref)
    finally:
        del os.environ["SSG_CHAIN_KBTREE"]


@pytest.mark.gpu
def test_gpu_every_chaining_form_follows_the_tree(gpu_lib, oracle, repeat_ref):
    reads = _constructed_reads(repeat_ref, 400, 4) + list(common.reads_with_inner_repeats(repeat_ref, 100, 10, rl=250))
    _check_forms(gpu_lib, oracle, repeat_ref, reads)
