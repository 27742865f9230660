"""CPU-side logic tests: the product's kernel sources, compiled against the host wave emulator
(tests/emu), must agree with the oracle bit for bit.  (The HIP build is tested by test_gpu_*.py.)"""
import pytest

import common
import numpy as np


def test_emu_extend(emu_lib, oracle):
    common.check_extend(emu_lib, oracle, 400, seed=1)


def test_emu_extend_lane_kernel(emu_lib, oracle, tmp_path):
    # row a7 as the product path runs it: ssg_k_ext_lane's ln_extend2 (the kernel behind mem_chain2aln), all three LDS classes
    assert common.check_extend_lane(emu_lib, oracle, 60, seed=61, workdir=tmp_path) == 720


def test_emu_seeds_sal(emu_lib, oracle, repeat_prefix):
    # row a3: ssg_k_sal's seed lists (bwt_sa + bns_intv2rid) directly, also on a repeat-rich reference and at the file's SA density
    assert common.check_seeds(emu_lib, oracle, 40, seed=62) > 1000
    assert common.check_seeds(emu_lib, oracle, 10, seed=63, prefix=repeat_prefix) > 1000


def test_emu_local(emu_lib, oracle):
    common.check_local(emu_lib, oracle, 60, seed=2)


def test_emu_local_lane_kernel(emu_lib, oracle, tmp_path):
    # row a10 as the product path runs it: forward pass by ssg_k_msw_lane (1 / 2 / 4 lanes per job), reverse pass by the wave code
    done, taken = common.check_local_lane(emu_lib, oracle, 40, seed=71, workdir=tmp_path)
    assert done == 240 and taken > 200
    done, taken = common.check_local_lane(emu_lib, oracle, 24, seed=72, workdir=tmp_path, lanes=(4,), scores=(2, 5, 7, 2, 9, 1))
    assert done == 48 and taken > 40


def test_emu_global(emu_lib, oracle):
    common.check_global(emu_lib, oracle, 150, seed=3)


def test_emu_smem(emu_lib, oracle):
    common.check_smem(emu_lib, oracle, 150, seed=4)


def test_emu_align1_150(emu_lib, oracle):
    assert common.check_align1(emu_lib, oracle, 200, seed=5) > 200


def test_emu_pe_sam_150(emu_lib, oracle):
    text, stats = common.check_pe_sam(emu_lib, oracle, 250, seed=7)
    assert text.count("\n") >= 500 and "SA:Z:" in text and "XA:Z:" in text


def test_emu_pe_sam_250_long_insert(emu_lib, oracle):
    # BASELINE.json config 5: 2x250, long inserts (wider SW band, more rescue / discordant content)
    text, stats = common.check_pe_sam(emu_lib, oracle, 60, seed=8, read_len=250, ins_mean=800, ins_std=150)
    assert text.count("\n") >= 120


def test_emu_pe_edge_cases(emu_lib, oracle):
    text = common.check_pe_edge_cases(emu_lib, oracle)
    flags = [int(l.split("\t")[1]) for l in text.split("\n") if l]
    assert any(f & 4 for f in flags) and any(f & 0x800 for f in flags)


def test_abi_argument_errors(emu_lib):
    import numpy as np
    import pytest
    from speedseq_amd import capi
    opt = emu_lib.opt_init()
    idx = emu_lib.index_load(common.EXAMPLE_FA)
    with pytest.raises(capi.SsgError):           # no pairs
        capi.mem_process_pairs(emu_lib, idx, opt, np.zeros(0, np.uint8), np.zeros(1, np.int64))
    long_read = np.zeros(600, dtype=np.uint8)     # reads beyond the supported length are refused, not truncated
    with pytest.raises(capi.SsgError):
        capi.mem_process_pairs(emu_lib, idx, opt, long_read, np.array([0, 300, 600], dtype=np.int64))
    with pytest.raises(capi.SsgError):           # missing index files
        emu_lib.index_load(common.EXAMPLE_FA + ".nope")
    emu_lib.index_destroy(idx)


def test_emu_dedup(emu_lib, oracle):
    assert common.check_dedup(emu_lib, oracle, 400, seed=9) > 20


def test_emu_align1_250(emu_lib, oracle):
    assert common.check_align1(emu_lib, oracle, 60, seed=6, read_len=250) > 60


def test_emu_repeats_align1(emu_lib, oracle, repeat_prefix, monkeypatch):
    # repeat-rich reference: reads with hundreds of seeds go through the wave-per-read chaining kernels
    assert common.check_align1(emu_lib, oracle, 12, seed=21, prefix=repeat_prefix) > 500
    monkeypatch.setenv("SSG_CHAIN_WAVE_BIG", "100")   # force the 4096-chain LDS variant
    monkeypatch.setenv("SSG_CHAIN_WAVE_MIN", "4")
    assert common.check_align1(emu_lib, oracle, 12, seed=22, prefix=repeat_prefix) > 500
    monkeypatch.setenv("SSG_CHAIN_RANKED", "0")         # the array-shifting insertion instead of the position-rank bitmap
    monkeypatch.setenv("SSG_CHAIN_WSORT", "0")          # and the weight sort on one lane,
    monkeypatch.setenv("SSG_CHAIN_BFLT", "0")           # the filter one chain at a time
    monkeypatch.setenv("SSG_CHAIN_SPEC", "0")           # (and, below with the bitmap again, the insertion seed by seed instead of 64 seeds a round)
    common.check_align1(emu_lib, oracle, 12, seed=22, prefix=repeat_prefix)
    monkeypatch.delenv("SSG_CHAIN_RANKED")
    monkeypatch.delenv("SSG_CHAIN_WSORT")
    monkeypatch.delenv("SSG_CHAIN_BFLT")
    monkeypatch.setenv("SSG_CHAIN_CAP_TEST", "40")      # the ranked form gives up at 40 chains: its fall-back, the shifting form, redoes those reads
    common.check_align1(emu_lib, oracle, 12, seed=22, prefix=repeat_prefix)
    monkeypatch.delenv("SSG_CHAIN_SPEC")
    common.check_align1(emu_lib, oracle, 12, seed=22, prefix=repeat_prefix)   # the same give-up out of a round of 64 seeds
    monkeypatch.delenv("SSG_CHAIN_CAP_TEST")
    monkeypatch.setenv("SSG_CHAIN_WAVE_MIN", "100000")  # and the lane-per-read kernel on the same reads
    monkeypatch.setenv("SSG_CHAIN_WAVE_BIG", "100000")
    assert common.check_align1(emu_lib, oracle, 12, seed=22, prefix=repeat_prefix) > 500


def test_emu_chain_weight_sort_by_the_wave(emu_lib):
    # upstream's unstable introsort of the chain weights (ties decide what the filter keeps): the wave replay against one lane running the textbook loops
    import numpy as np
    rng = np.random.default_rng(3)
    for n in list(range(0, 40)) + [64, 65, 127, 129, 300, 1024, 1025, 2500]:
        for kind in range(6):
            w = [rng.integers(0, 150, n), rng.integers(0, 4, n), np.full(n, 17), np.sort(rng.integers(0, 50, n)), np.sort(rng.integers(0, 50, n))[::-1].copy(),
                 np.where(rng.random(n) < 0.9, 19, rng.integers(0, 40, n))][kind]
            keys = (w.astype(np.int64) << 32) | np.arange(n, dtype=np.int64)
            a, b = emu_lib.dbg_chain_sort(keys)
            assert np.array_equal(a, b), (n, kind)
            assert np.all(np.diff(a >> 32) <= 0) and np.array_equal(np.sort(a), np.sort(keys))
    for n in (17, 100, 600, 1200, 5120):   # distinct keys in order, organ pipes, interleavings: upstream's pivot rule runs out of depth on these and switches to combsort
        for w in (np.arange(n), np.arange(n)[::-1].copy(), np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]), np.concatenate([np.arange(n // 2) * 2, np.arange(n - n // 2) * 2 + 1]),
                  np.ravel(np.column_stack([np.arange(n // 2), np.arange(n // 2)[::-1]])), np.arange(n) // 3):
            keys = (w.astype(np.int64) << 32) | np.arange(len(w), dtype=np.int64)
            a, b = emu_lib.dbg_chain_sort(keys)
            assert np.array_equal(a, b), n
            assert np.all(np.diff(a >> 32) <= 0)


def test_emu_chain_filter_options(emu_lib, oracle, repeat_mid_prefix, monkeypatch):
    # mem_chain_flt's knobs away from their defaults (drop ratio, mask level, minimum weight, the cap on extended chains, the gap that makes an overlap count):
    # the light reads' lane kernel, then the same reads through the wave kernels
    sets = [(0.3, 0.2, 60, 10, 40), (0.8, 0.5, 0, 1 << 30, 100), (0.95, 0.9, 30, 3, 10000), (0.5, 0.5, 0, 50, 40)]
    for wmin in ("64", "4"):
        monkeypatch.setenv("SSG_CHAIN_WAVE_MIN", wmin)
        for k, co in enumerate(sets):
            assert common.check_align1(emu_lib, oracle, 60, seed=40 + k, read_len=(150, 250)[k & 1], prefix=repeat_mid_prefix, chain_opt=co) > 0


def test_emu_extension_column_classes(emu_lib, oracle, monkeypatch):
    # the lane-per-extension kernel with the LDS its class's longest side needs (classes of 8 columns, three queues), then the fixed classes (72 / 136 / 256 / 320 columns)
    for dyn in ("1", "0"):
        monkeypatch.setenv("SSG_EXT_DYN", dyn)
        for k, rl in enumerate((150, 250, 300, 101)):
            assert common.check_align1(emu_lib, oracle, 40, seed=50 + k, read_len=rl) > 40


def test_emu_chains_at_equal_positions(emu_lib, oracle, monkeypatch):
    # reads with the same reference segment two or three times, too far apart to merge: a second chain at one position (upstream's order among equal positions), a third
    # (the ranked wave form gives the read up and redoes it) -- through the lane kernels (LDS, global) and every form of the wave kernels
    seqs = common.reads_with_inner_repeats(common.EXAMPLE_FA, 120, 5) + common.reads_with_inner_repeats(common.EXAMPLE_FA, 80, 6, rl=150)
    counts = set()
    for env in ({}, {"SSG_CHAIN_LDS": "0"}, {"SSG_CHAIN_WAVE_MIN": "1"}, {"SSG_CHAIN_WAVE_MIN": "1", "SSG_CHAIN_SPEC": "0"}, {"SSG_CHAIN_WAVE_MIN": "1", "SSG_CHAIN_RANKED": "0"},
                {"SSG_CHAIN_WAVE_MIN": "1", "SSG_CHAIN_BFLT": "0", "SSG_CHAIN_WSORT": "0"}):
        for k in ("SSG_CHAIN_LDS", "SSG_CHAIN_WAVE_MIN", "SSG_CHAIN_SPEC", "SSG_CHAIN_RANKED", "SSG_CHAIN_BFLT", "SSG_CHAIN_WSORT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        counts.add(common.check_align1_reads(emu_lib, oracle, seqs))
    assert len(counts) == 1 and min(counts) > 120


def test_emu_light_reads_chain_lds(emu_lib, oracle, repeat_mid_prefix, monkeypatch):
    # reads with 10..63 seeds in small repeat families: the three classes of ssg_k_chain_lds (state in the lane's LDS), then the same reads through ssg_k_chain
    monkeypatch.setenv("SSG_CHAIN_WAVE_MIN", "64")
    n1 = common.check_align1(emu_lib, oracle, 150, seed=31, prefix=repeat_mid_prefix)
    n2 = common.check_align1(emu_lib, oracle, 100, seed=32, read_len=250, prefix=repeat_mid_prefix)
    assert n1 > 600 and n2 > 400
    monkeypatch.setenv("SSG_CHAIN_LDS", "0")
    assert common.check_align1(emu_lib, oracle, 150, seed=31, prefix=repeat_mid_prefix) == n1


def test_emu_repeats_pe_sam(emu_lib, oracle, repeat_prefix):
    text, stats = common.check_pe_sam(emu_lib, oracle, 10, seed=23, prefix=repeat_prefix)
    assert "XA:Z:" in text


def test_emu_pe_sam_300(emu_lib, oracle, tmp_path, repeat_prefix):
    # 2x300 (the reference's script takes any read length, bin/speedseq:196-200): the column classes above 256 -- fifth register column of the
    # wave SW (k_sw.h), ssg_k_ext_lane<320>, 9-bit column tags of the lane kernels, 16-bit query coordinates of the wave chaining, 40 LDS words of
    # the seeding kernels -- end to end against the oracle, and the stage twins at those lengths
    text, stats = common.check_pe_sam(emu_lib, oracle, 250, seed=19, read_len=300, ins_mean=900, ins_std=150)
    assert text.count("\n") >= 500
    assert common.check_align1(emu_lib, oracle, 200, seed=20, read_len=300) > 400
    assert common.check_extend_lane(emu_lib, oracle, 60, seed=9, workdir=tmp_path, qcaps=(320,)) == 240
    common.check_extend(emu_lib, oracle, 150, seed=21, max_qlen=318)
    done, taken = common.check_local_lane(emu_lib, oracle, 24, seed=22, workdir=tmp_path, lanes=(4, 2), qlens=(300, 300, 310, 280, 257, 264))
    assert done == 96 and taken > 80
    assert common.check_align1(emu_lib, oracle, 40, seed=24, read_len=300, prefix=repeat_prefix) > 1000   # wave-per-read chaining: query coordinates beyond 255


def test_emu_repeats_mate_rescue(emu_lib, oracle, repeat_pe_prefix):
    text, stats = common.check_pe_sam(emu_lib, oracle, 250, seed=5, prefix=repeat_pe_prefix)
    assert stats[3] > 1000   # rescues


def test_emu_pair_wave_kernel_forced(emu_lib, oracle, repeat_pe_prefix, monkeypatch):
    # every pair through the wave-per-pair primary-marking / pairing kernel (normally only long region lists)
    monkeypatch.setenv("SSG_PAIR_WAVE_MIN", "0")
    common.check_pe_sam(emu_lib, oracle, 250, seed=7)
    common.check_pe_edge_cases(emu_lib, oracle)
    common.check_pe_sam(emu_lib, oracle, 120, seed=8, prefix=repeat_pe_prefix)


def test_emu_smem_budget_and_wave_kernel(emu_lib, oracle, repeat_prefix, monkeypatch):
    # extension budget of the lane kernel: given-up reads are redone by the wave-per-read kernel (k_smem2.h)
    monkeypatch.setenv("SSG_SMEM_MAX_EXT", "1")        # every read given up at once: the wave kernel does them all, both list classes
    common.check_smem(emu_lib, oracle, 60, seed=41)
    common.check_smem(emu_lib, oracle, 30, seed=42, read_len=250)
    common.check_smem(emu_lib, oracle, 40, seed=44, n_frac=0.02)
    common.check_smem(emu_lib, oracle, 8, seed=43, prefix=repeat_prefix, cap=512)
    monkeypatch.setenv("SSG_SMEM_MAX_EXT", "700")      # some of each
    monkeypatch.setenv("SSG_SMEM_MAX_ROW", "18")       # ... and reads whose first row is longer than this
    common.check_smem(emu_lib, oracle, 60, seed=41)
    common.check_smem(emu_lib, oracle, 8, seed=43, prefix=repeat_prefix, cap=512)
    monkeypatch.setenv("SSG_SMEM_MAX_EXT", "2147483647")   # no budget: the lane kernel alone, also on ambiguous bases and repeats
    common.check_smem(emu_lib, oracle, 40, seed=44, n_frac=0.02)
    common.check_smem(emu_lib, oracle, 8, seed=43, prefix=repeat_prefix, cap=512)


def test_emu_smem_table_of_short_pattern_intervals(emu_lib, oracle, repeat_prefix, monkeypatch):
    # the table of the intervals of all patterns up to K bases (k_smem2.h): a bwt_extend whose result is that short is one load, the third pass
    # starts K bases in; K is chosen at index load.  Same intervals for every K, with ambiguous bases, on repeats, with give-ups, and without a table
    for k in ("2", "5", "11", "0", "25"):              # 25: capped by what the index is worth (log4 of the text + 2) and by min_seed_len - 1 in the kernel
        monkeypatch.setenv("SSG_KTAB_K", k)
        monkeypatch.setenv("SSG_KTAB_VERIFY", "1")
        common.check_smem(emu_lib, oracle, 60, seed=51)
        common.check_smem(emu_lib, oracle, 60, seed=52, n_frac=0.03)
    monkeypatch.setenv("SSG_KTAB_K", "9")
    common.check_smem(emu_lib, oracle, 8, seed=43, prefix=repeat_prefix, cap=512)
    monkeypatch.setenv("SSG_SMEM_MAX_EXT", "300")
    common.check_smem(emu_lib, oracle, 60, seed=53, n_frac=0.01)
    monkeypatch.delenv("SSG_SMEM_MAX_EXT")
    monkeypatch.setenv("SSG_SMEM_USE_KTAB", "0")       # a table in the index, not used
    common.check_smem(emu_lib, oracle, 60, seed=51)


def test_emu_smem_kernel_variants(emu_lib, oracle, repeat_prefix, monkeypatch):
    monkeypatch.setenv("SSG_SMEM_KERNEL", "lane")     # the nested-loop form (the product kernels' fall-back), on the same reads
    common.check_smem(emu_lib, oracle, 150, seed=31)
    monkeypatch.delenv("SSG_SMEM_KERNEL")
    monkeypatch.setenv("SSG_SA_INTV", "32")
    assert common.check_align1(emu_lib, oracle, 150, seed=33) > 150
    monkeypatch.delenv("SSG_SA_INTV")
    # the forms the round's last kernels replaced stay behind switches: introsort by a lane per read; the locate stage's walks instead of running counts / running maximum
    monkeypatch.setenv("SSG_SMEM_SORT_RANK", "0")
    common.check_smem(emu_lib, oracle, 60, seed=35)
    common.check_smem(emu_lib, oracle, 8, seed=43, prefix=repeat_prefix, cap=512)
    monkeypatch.delenv("SSG_SMEM_SORT_RANK")
    common.check_smem(emu_lib, oracle, 8, seed=43, prefix=repeat_prefix, cap=512)   # lists beyond 24 intervals: the wave form of the sort
    monkeypatch.setenv("SSG_SAL_PREFIX", "0")
    monkeypatch.setenv("SSG_SAL_READ_OF", "0")
    assert common.check_align1(emu_lib, oracle, 150, seed=36) > 150


@pytest.mark.parametrize("read_len", [150, 250])
def test_emu_hotpath_dedup_and_classification(emu_lib, oracle, read_len):
    """the bench's step through ssg_hotpath_dev_ex (rows a1-a17 in one call, samblaster's decisions on the device records) vs the oracle"""
    r = common.check_hotpath(emu_lib, oracle, 600, 200, read_len, lambda a: (a, a.ctypes.data))
    assert r[0] >= 1200


def test_emu_owner_side_dedup_matches_min_ordinal_rule(emu_lib):
    """ssg_markdup_sig_dev (the owner rank's side of the signature exchange, SURVEY 8e coupling 2): an element is a duplicate iff the same
    signature came with a smaller global ordinal; all-ones signatures never are -- against the rule written out in numpy"""
    import numpy as np
    from speedseq_amd import capi
    rng = np.random.default_rng(7)
    n = 5000
    sig = rng.integers(0, 6, size=(n, 3)).astype(np.uint64)           # many collisions
    sig[rng.random(n) < 0.1] = np.uint64(0xffffffffffffffff)           # never-duplicate entries
    ordinal = rng.permutation(n * 3)[:n].astype(np.int64)             # arrival order is not input order
    dup = np.zeros(n, dtype=np.uint8)
    capi.markdup_sig_dev(emu_lib, n, sig.ctypes.data, ordinal.ctypes.data, dup.ctypes.data)
    first = {}
    for i in np.argsort(ordinal):
        k = tuple(int(x) for x in sig[i])
        first.setdefault(k, int(ordinal[i]))
    never = (0xffffffffffffffff,) * 3
    exp = np.array([0 if tuple(int(x) for x in sig[i]) == never else int(ordinal[i] > first[tuple(int(x) for x in sig[i])]) for i in range(n)], dtype=np.uint8)
    assert np.array_equal(dup, exp) and exp.sum() > n // 2


def test_emu_reg2aln_lane_dp_classes(emu_lib, oracle, monkeypatch):
    # ksw_global2 + backtrace + NM / MD with one lane per record (k_aln.h ssg_k_reg2aln_dplane: ring of 2w + 2 columns in LDS, 4-bit direction cells in
    # an HBM slab) against the oracle, at error and indel rates that fill all three band classes and leave records to the wave kernel; then with the
    # lane DP switched off: the same text both ways (check_pe_sam compares with the oracle's SAM line by line)
    import ctypes as C
    seen = []
    for k, (err, indel, rl) in enumerate(((0.02, 0.01, 150), (0.05, 0.004, 150), (0.03, 0.01, 250), (0.01, 0.02, 101))):
        texts = []
        for dp in ("1", "0"):
            monkeypatch.setenv("SSG_R2A_DPLANE", dp)
            text, _ = common.check_pe_sam(emu_lib, oracle, 120, seed=300 + k, read_len=rl, err=err, indel_frac=indel, ins_mean=500 if rl < 250 else 800, ins_std=60)
            texts.append(text)
            if dp == "1":
                c = (C.c_uint * 4)()
                emu_lib.l.ssg_dbg_reg2aln_counts(c)
                seen.append([c[1], c[2], c[3], c[0]])
        assert texts[0] == texts[1]
    tot = np.array(seen).sum(axis=0)
    assert all(tot[:3] > 0) and tot[3] > 0, seen   # every class of the lane DP and the wave kernel had records


@pytest.mark.parametrize("refill", ["1", "32", "64"])
def test_emu_sa_densify_walks_refilled_in_batches(emu_lib, monkeypatch, refill):
    # the denser suffix-array copy (k_seed.h ssg_k_sa_densify_walk): idle lanes take new walks when `refill` of them wait; every new sample checked against bwt_sa (SSG_SA_VERIFY)
    monkeypatch.setenv("SSG_SA_VERIFY", "1")
    monkeypatch.setenv("SSG_DENSIFY_REFILL", refill)
    for intv in ("4", "8", "1"):
        monkeypatch.setenv("SSG_SA_INTV", intv)
        idx = emu_lib.index_load(common.EXAMPLE_FA)
        emu_lib.index_destroy(idx)


def test_emu_index_densify_to_steps(emu_lib, oracle):
    # ssg_index_load2 with the file's samples, then ssg_index_densify_to in two steps (16, then 4; a step that is not denser is a no-op): the same alignments as the oracle's
    import ctypes as C
    h = C.c_void_p()
    assert emu_lib.l.ssg_index_load2(common.EXAMPLE_FA.encode(), 1, C.byref(h)) == 0
    for intv in (16, 32, 4, 8):
        assert emu_lib.l.ssg_index_densify_to(h, C.c_int(intv)) == 0
    assert emu_lib.l.ssg_index_densify_to(h, C.c_int(0)) != 0
    emu_lib.index_destroy(h)
    assert common.check_align1(emu_lib, oracle, 60, seed=77) > 60
