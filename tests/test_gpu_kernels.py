"""GPU parity tests (HIP build on a real MI355X) against the oracle, through the C ABI."""
import pytest

import common

pytestmark = pytest.mark.gpu


def test_gpu_backend(gpu_lib):
    assert gpu_lib.backend().startswith("hip")


def test_gpu_extend(gpu_lib, oracle):
    common.check_extend(gpu_lib, oracle, 3000, seed=11)


def test_gpu_extend_lane_kernel(gpu_lib, oracle, tmp_path):
    # row a7 as the product path runs it: ssg_k_ext_lane's ln_extend2 (the kernel behind mem_chain2aln), all three LDS classes
    assert common.check_extend_lane(gpu_lib, oracle, 1000, seed=61, workdir=tmp_path) == 12000


def test_gpu_seeds_sal(gpu_lib, oracle, repeat_prefix):
    # row a3: ssg_k_sal's seed lists (bwt_sa + bns_intv2rid) directly, also on a repeat-rich reference and at the file's SA density
    assert common.check_seeds(gpu_lib, oracle, 1500, seed=62) > 1000
    assert common.check_seeds(gpu_lib, oracle, 10, seed=63, prefix=repeat_prefix) > 1000


def test_gpu_local(gpu_lib, oracle):
    common.check_local(gpu_lib, oracle, 400, seed=12)


def test_gpu_local_lane_kernel(gpu_lib, oracle, tmp_path):
    # row a10 as the product path runs it (k_mswlane.h): the forward pass of ksw_align2 by ssg_k_msw_lane with 1 / 2 / 4 lanes per job, the
    # reverse pass by the wave code; both strands of the 2-bit reference, N in queries, other scoring
    done, taken = common.check_local_lane(gpu_lib, oracle, 1500, seed=71, workdir=tmp_path)
    assert done == 9000 and taken > 8000
    done, taken = common.check_local_lane(gpu_lib, oracle, 600, seed=72, workdir=tmp_path, scores=(2, 5, 7, 2, 9, 1))
    assert done == 3600 and taken > 3000
    done, taken = common.check_local_lane(gpu_lib, oracle, 300, seed=73, workdir=tmp_path, lanes=(4,), scores=(1, 4, 6, 1, 6, 1))
    assert done == 600


def test_gpu_global(gpu_lib, oracle):
    common.check_global(gpu_lib, oracle, 1000, seed=13)


def test_gpu_smem(gpu_lib, oracle):
    common.check_smem(gpu_lib, oracle, 2000, seed=14)


def test_gpu_align1_150(gpu_lib, oracle):
    assert common.check_align1(gpu_lib, oracle, 4000, seed=15) > 4000


def test_gpu_pe_sam_150(gpu_lib, oracle):
    text, stats = common.check_pe_sam(gpu_lib, oracle, 5000, seed=17)
    assert text.count("\n") >= 10000 and "SA:Z:" in text and "XA:Z:" in text
    assert stats[3] > 0  # some mate rescues happened


def test_gpu_pe_sam_250_long_insert(gpu_lib, oracle):
    text, stats = common.check_pe_sam(gpu_lib, oracle, 1500, seed=18, read_len=250, ins_mean=800, ins_std=150)
    assert text.count("\n") >= 3000


def test_gpu_pe_sam_300(gpu_lib, oracle, tmp_path, repeat_prefix):
    # 2x300: every column class above 256 (see test_emu_pe_sam_300), end to end and stage by stage
    text, stats = common.check_pe_sam(gpu_lib, oracle, 1200, seed=19, read_len=300, ins_mean=900, ins_std=150)
    assert text.count("\n") >= 2400
    assert common.check_align1(gpu_lib, oracle, 1500, seed=20, read_len=300) > 3000
    assert common.check_extend_lane(gpu_lib, oracle, 500, seed=9, workdir=tmp_path, qcaps=(320,)) == 2000
    common.check_extend(gpu_lib, oracle, 1500, seed=21, max_qlen=318)
    done, taken = common.check_local_lane(gpu_lib, oracle, 400, seed=22, workdir=tmp_path, lanes=(4, 2, 1), qlens=(300, 300, 310, 280, 257, 264))
    assert done == 2400 and taken > 2000
    common.check_smem(gpu_lib, oracle, 600, seed=23, read_len=300, cap=192)
    assert common.check_align1(gpu_lib, oracle, 200, seed=24, read_len=300, prefix=repeat_prefix) > 5000   # wave-per-read chaining: query coordinates beyond 255


def test_gpu_pe_edge_cases(gpu_lib, oracle):
    common.check_pe_edge_cases(gpu_lib, oracle)


def test_gpu_pe_sam_multibatch_properties(gpu_lib, oracle):
    """Larger run (no oracle diff): size-independent properties of the records -- every read present,
    flags consistent, mate fields symmetric, duplicate marking idempotent under re-ordering of nothing."""
    import numpy as np
    from speedseq_amd import capi
    gidx = gpu_lib.index_load(common.EXAMPLE_FA)
    pairs, seqs, seq, off = common.sim_reads(30000, 41)
    pb = (np.arange(30000) // 10000).astype(np.int32)
    opt = gpu_lib.opt_init()
    res = capi.mem_process_pairs(gpu_lib, gidx, opt, seq, off, pair_batch=pb, n_batches=3)
    assert len(res.req_off) == 60001 and np.all(np.diff(res.req_off) >= 1)
    main0 = res.alns[res.req_off[:-1]]
    assert np.all((main0["flag"] & 0x900) == 0)
    assert np.all(res.req["kind"][res.req_off[:-1]] == 0)
    mapped = main0["rid"] >= 0
    assert mapped.mean() > 0.95
    assert np.all(main0["n_cigar"][mapped] > 0) and np.all(main0["mapq"] <= 60)
    for b in range(3):
        assert res.pes["failed"][4 * b + 1] == 0 and 350 < res.pes["avg"][4 * b + 1] < 450
    res.close()
    gpu_lib.index_destroy(gidx)


def test_gpu_dedup(gpu_lib, oracle):
    assert common.check_dedup(gpu_lib, oracle, 20000, seed=19) > 1000


def test_gpu_align1_250(gpu_lib, oracle):
    assert common.check_align1(gpu_lib, oracle, 1000, seed=16, read_len=250) > 1000


def test_gpu_repeats_align1(gpu_lib, oracle, repeat_prefix, monkeypatch):
    # repeat-rich reference: reads with hundreds to thousands of seeds (wave-per-read chaining kernels)
    assert common.check_align1(gpu_lib, oracle, 300, seed=21, prefix=repeat_prefix) > 10000
    monkeypatch.setenv("SSG_CHAIN_WAVE_BIG", "100")   # force the 4096-chain LDS variant
    monkeypatch.setenv("SSG_CHAIN_WAVE_MIN", "4")
    assert common.check_align1(gpu_lib, oracle, 150, seed=22, prefix=repeat_prefix) > 5000
    monkeypatch.setenv("SSG_CHAIN_RANKED", "0")         # the array-shifting insertion instead of the position-rank bitmap
    monkeypatch.setenv("SSG_CHAIN_WSORT", "0")          # and the weight sort on one lane,
    monkeypatch.setenv("SSG_CHAIN_BFLT", "0")           # the filter one chain at a time
    monkeypatch.setenv("SSG_CHAIN_SPEC", "0")           # (and, below with the bitmap again, the insertion seed by seed instead of 64 seeds a round)
    common.check_align1(gpu_lib, oracle, 300, seed=22, prefix=repeat_prefix)
    monkeypatch.delenv("SSG_CHAIN_RANKED")
    monkeypatch.delenv("SSG_CHAIN_WSORT")
    monkeypatch.delenv("SSG_CHAIN_BFLT")
    monkeypatch.setenv("SSG_CHAIN_CAP_TEST", "40")      # the ranked form gives up at 40 chains: its fall-back, the shifting form, redoes those reads
    common.check_align1(gpu_lib, oracle, 300, seed=22, prefix=repeat_prefix)
    monkeypatch.delenv("SSG_CHAIN_SPEC")
    common.check_align1(gpu_lib, oracle, 300, seed=22, prefix=repeat_prefix)   # the same give-up out of a round of 64 seeds
    monkeypatch.delenv("SSG_CHAIN_CAP_TEST")
    monkeypatch.setenv("SSG_CHAIN_WAVE_MIN", "100000")  # and the lane-per-read kernel on the same reads
    monkeypatch.setenv("SSG_CHAIN_WAVE_BIG", "100000")
    assert common.check_align1(gpu_lib, oracle, 150, seed=22, prefix=repeat_prefix) > 5000


def test_gpu_chain_weight_sort_by_the_wave(gpu_lib):
    # upstream's unstable introsort of the chain weights (ties decide what the filter keeps): the wave replay against one lane running the textbook loops
    import numpy as np
    rng = np.random.default_rng(3)
    for n in list(range(0, 70)) + [100, 127, 128, 129, 255, 256, 257, 500, 1000, 1023, 1024, 1025, 2000, 3000, 4000, 5120]:
        for kind in range(6):
            w = [rng.integers(0, 150, n), rng.integers(0, 4, n), np.full(n, 17), np.sort(rng.integers(0, 50, n)), np.sort(rng.integers(0, 50, n))[::-1].copy(),
                 np.where(rng.random(n) < 0.9, 19, rng.integers(0, 40, n))][kind]
            keys = (w.astype(np.int64) << 32) | np.arange(n, dtype=np.int64)
            a, b = gpu_lib.dbg_chain_sort(keys)
            assert np.array_equal(a, b), (n, kind)
            assert np.all(np.diff(a >> 32) <= 0) and np.array_equal(np.sort(a), np.sort(keys))
    for n in (17, 100, 600, 1200, 5120):   # distinct keys in order, organ pipes, interleavings: upstream's pivot rule runs out of depth on these and switches to combsort
        for w in (np.arange(n), np.arange(n)[::-1].copy(), np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]), np.concatenate([np.arange(n // 2) * 2, np.arange(n - n // 2) * 2 + 1]),
                  np.ravel(np.column_stack([np.arange(n // 2), np.arange(n // 2)[::-1]])), np.arange(n) // 3):
            keys = (w.astype(np.int64) << 32) | np.arange(len(w), dtype=np.int64)
            a, b = gpu_lib.dbg_chain_sort(keys)
            assert np.array_equal(a, b), n
            assert np.all(np.diff(a >> 32) <= 0)


def test_gpu_chain_filter_options(gpu_lib, oracle, repeat_mid_prefix, monkeypatch):
    # mem_chain_flt's knobs away from their defaults (drop ratio, mask level, minimum weight, the cap on extended chains, the gap that makes an overlap count):
    # the light reads' lane kernel, then the same reads through the wave kernels
    sets = [(0.3, 0.2, 60, 10, 40), (0.8, 0.5, 0, 1 << 30, 100), (0.95, 0.9, 30, 3, 10000), (0.5, 0.5, 0, 50, 40)]
    for wmin in ("64", "4"):
        monkeypatch.setenv("SSG_CHAIN_WAVE_MIN", wmin)
        for k, co in enumerate(sets):
            assert common.check_align1(gpu_lib, oracle, 600, seed=40 + k, read_len=(150, 250)[k & 1], prefix=repeat_mid_prefix, chain_opt=co) > 0


def test_gpu_extension_column_classes(gpu_lib, oracle, monkeypatch):
    # the lane-per-extension kernel with the LDS its class's longest side needs (classes of 8 columns, three queues), then the fixed classes (72 / 136 / 256 / 320 columns)
    for dyn in ("1", "0"):
        monkeypatch.setenv("SSG_EXT_DYN", dyn)
        for k, rl in enumerate((150, 250, 300, 101)):
            assert common.check_align1(gpu_lib, oracle, 1500, seed=50 + k, read_len=rl) > 1500


def test_gpu_chains_at_equal_positions(gpu_lib, oracle, monkeypatch):
    # reads with the same reference segment two or three times, too far apart to merge: a second chain at one position (upstream's order among equal positions), a third
    # (the ranked wave form gives the read up and redoes it) -- through the lane kernels (LDS, global) and every form of the wave kernels
    seqs = common.reads_with_inner_repeats(common.EXAMPLE_FA, 1500, 5) + common.reads_with_inner_repeats(common.EXAMPLE_FA, 1000, 6, rl=150)
    counts = set()
    for env in ({}, {"SSG_CHAIN_LDS": "0"}, {"SSG_CHAIN_WAVE_MIN": "1"}, {"SSG_CHAIN_WAVE_MIN": "1", "SSG_CHAIN_SPEC": "0"}, {"SSG_CHAIN_WAVE_MIN": "1", "SSG_CHAIN_RANKED": "0"},
                {"SSG_CHAIN_WAVE_MIN": "1", "SSG_CHAIN_BFLT": "0", "SSG_CHAIN_WSORT": "0"}):
        for k in ("SSG_CHAIN_LDS", "SSG_CHAIN_WAVE_MIN", "SSG_CHAIN_SPEC", "SSG_CHAIN_RANKED", "SSG_CHAIN_BFLT", "SSG_CHAIN_WSORT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        counts.add(common.check_align1_reads(gpu_lib, oracle, seqs))
    assert len(counts) == 1 and min(counts) > 1500


def test_gpu_light_reads_chain_lds(gpu_lib, oracle, repeat_mid_prefix, monkeypatch):
    # reads with 10..63 seeds in small repeat families: the three classes of ssg_k_chain_lds (state in the lane's LDS), then the same reads through ssg_k_chain
    monkeypatch.setenv("SSG_CHAIN_WAVE_MIN", "64")
    n1 = common.check_align1(gpu_lib, oracle, 1500, seed=31, prefix=repeat_mid_prefix)
    n2 = common.check_align1(gpu_lib, oracle, 1000, seed=32, read_len=250, prefix=repeat_mid_prefix)
    assert n1 > 6000 and n2 > 4000
    monkeypatch.setenv("SSG_CHAIN_LDS", "0")
    assert common.check_align1(gpu_lib, oracle, 1500, seed=31, prefix=repeat_mid_prefix) == n1


def test_gpu_repeats_pe_sam(gpu_lib, oracle, repeat_prefix):
    text, stats = common.check_pe_sam(gpu_lib, oracle, 300, seed=23, prefix=repeat_prefix)
    assert "XA:Z:" in text


def test_gpu_repeats_mate_rescue(gpu_lib, oracle, repeat_pe_prefix):
    text, stats = common.check_pe_sam(gpu_lib, oracle, 1500, seed=5, prefix=repeat_pe_prefix)
    assert stats[3] > 10000   # rescues


def test_gpu_pair_wave_kernel_forced(gpu_lib, oracle, repeat_pe_prefix, monkeypatch):
    # every pair through the wave-per-pair primary-marking / pairing kernel (normally only long region lists)
    monkeypatch.setenv("SSG_PAIR_WAVE_MIN", "0")
    common.check_pe_sam(gpu_lib, oracle, 3000, seed=7)
    common.check_pe_edge_cases(gpu_lib, oracle)
    common.check_pe_sam(gpu_lib, oracle, 600, seed=8, prefix=repeat_pe_prefix)


def test_gpu_smem_budget_and_wave_kernel(gpu_lib, oracle, repeat_prefix, monkeypatch):
    # extension budget of the lane kernel: given-up reads are redone by the wave-per-read kernel (k_smem2.h)
    monkeypatch.setenv("SSG_SMEM_MAX_EXT", "1")        # every read given up at once: the wave kernel does them all, both list classes
    common.check_smem(gpu_lib, oracle, 1500, seed=41)
    common.check_smem(gpu_lib, oracle, 500, seed=42, read_len=250)
    common.check_smem(gpu_lib, oracle, 500, seed=44, n_frac=0.02)
    common.check_smem(gpu_lib, oracle, 40, seed=43, prefix=repeat_prefix, cap=512)
    monkeypatch.setenv("SSG_SMEM_MAX_EXT", "700")      # some of each
    monkeypatch.setenv("SSG_SMEM_MAX_ROW", "18")       # ... and reads whose first row is longer than this
    common.check_smem(gpu_lib, oracle, 1500, seed=41)
    common.check_smem(gpu_lib, oracle, 40, seed=43, prefix=repeat_prefix, cap=512)
    monkeypatch.setenv("SSG_SMEM_MAX_EXT", "2147483647")   # no budget: the lane kernel alone, also on ambiguous bases and repeats
    common.check_smem(gpu_lib, oracle, 500, seed=44, n_frac=0.02)
    common.check_smem(gpu_lib, oracle, 40, seed=43, prefix=repeat_prefix, cap=512)


def test_gpu_smem_table_of_short_pattern_intervals(gpu_lib, oracle, repeat_prefix, monkeypatch):
    # the table of the intervals of all patterns up to K bases (k_smem2.h): a bwt_extend whose result is that short is one load, the third pass
    # starts K bases in; K is chosen at index load.  Same intervals for every K, with ambiguous bases, on repeats, with give-ups, and without a table
    for k in ("2", "5", "11", "0", "25"):              # 25: capped by what the index is worth (log4 of the text + 2) and by min_seed_len - 1 in the kernel
        monkeypatch.setenv("SSG_KTAB_K", k)
        monkeypatch.setenv("SSG_KTAB_VERIFY", "1")
        common.check_smem(gpu_lib, oracle, 1500, seed=51)
        common.check_smem(gpu_lib, oracle, 1500, seed=52, n_frac=0.03)
    monkeypatch.setenv("SSG_KTAB_K", "9")
    common.check_smem(gpu_lib, oracle, 8, seed=43, prefix=repeat_prefix, cap=512)
    monkeypatch.setenv("SSG_SMEM_MAX_EXT", "300")
    common.check_smem(gpu_lib, oracle, 1500, seed=53, n_frac=0.01)
    monkeypatch.delenv("SSG_SMEM_MAX_EXT")
    monkeypatch.setenv("SSG_SMEM_USE_KTAB", "0")       # a table in the index, not used
    common.check_smem(gpu_lib, oracle, 1500, seed=51)


def test_gpu_smem_kernel_variants(gpu_lib, oracle, monkeypatch):
    monkeypatch.setenv("SSG_SMEM_KERNEL", "lane")     # the nested-loop form (the product kernels' fall-back), on the same reads
    common.check_smem(gpu_lib, oracle, 1500, seed=31)
    monkeypatch.delenv("SSG_SMEM_KERNEL")
    monkeypatch.setenv("SSG_SA_INTV", "32")   # the file's own suffix-array density
    assert common.check_align1(gpu_lib, oracle, 1500, seed=33) > 1500
    monkeypatch.delenv("SSG_SA_INTV")
    # the forms the round's last kernels replaced stay behind switches: introsort by a lane per read; the locate stage's walks instead of running counts / running maximum
    monkeypatch.setenv("SSG_SMEM_SORT_RANK", "0")
    common.check_smem(gpu_lib, oracle, 1500, seed=35)
    monkeypatch.delenv("SSG_SMEM_SORT_RANK")
    monkeypatch.setenv("SSG_SAL_PREFIX", "0")
    monkeypatch.setenv("SSG_SAL_READ_OF", "0")
    assert common.check_align1(gpu_lib, oracle, 1500, seed=36) > 1500


@pytest.mark.parametrize("read_len", [150, 250])
def test_gpu_hotpath_batches_and_dups(gpu_lib, read_len):
    """ssg_hotpath_dev_ex (the bench's step: alignment, duplicate marking, discordant / splitter classification on the device) on
    device-resident reads in three upstream batches against the oracle, at 2x150 and 2x250 with long inserts; runs in its own
    process (tests/hotpath_check.py) because torch must initialise its HIP runtime before libssgpu is loaded, as in bench.py."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "hotpath_check.py"), str(read_len)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "hotpath ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]

@pytest.mark.gpu
def test_gpu_reg2aln_lane_dp_classes(gpu_lib, oracle, monkeypatch):
    # MI355X twin of test_emu_reg2aln_lane_dp_classes: the lane-per-record banded global alignment (three band classes) and the wave kernel's share,
    # against the oracle's SAM text; more pairs than the emulation takes
    import ctypes as C
    import numpy as np
    seen = []
    for k, (err, indel, rl) in enumerate(((0.02, 0.01, 150), (0.05, 0.004, 150), (0.03, 0.01, 250), (0.01, 0.02, 101))):
        texts = []
        for dp in ("1", "0"):
            monkeypatch.setenv("SSG_R2A_DPLANE", dp)
            text, _ = common.check_pe_sam(gpu_lib, oracle, 400, seed=300 + k, read_len=rl, err=err, indel_frac=indel, ins_mean=500 if rl < 250 else 800, ins_std=60)
            texts.append(text)
            if dp == "1":
                c = (C.c_uint * 4)()
                gpu_lib.l.ssg_dbg_reg2aln_counts(c)
                seen.append([c[1], c[2], c[3], c[0]])
        assert texts[0] == texts[1]
    tot = np.array(seen).sum(axis=0)
    assert all(tot[:3] > 0) and tot[3] > 0, seen
