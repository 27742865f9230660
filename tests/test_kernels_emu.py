"""CPU-side logic tests: the product's kernel sources, compiled against the host wave emulator
(tests/emu), must agree with the oracle bit for bit.  (The HIP build is tested by test_gpu_*.py.)"""
import common


def test_emu_extend(emu_lib, oracle):
    common.check_extend(emu_lib, oracle, 400, seed=1)


def test_emu_local(emu_lib, oracle):
    common.check_local(emu_lib, oracle, 60, seed=2)


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


def test_emu_dedup(emu_lib, oracle):
    assert common.check_dedup(emu_lib, oracle, 400, seed=9) > 20


def test_emu_align1_250(emu_lib, oracle):
    assert common.check_align1(emu_lib, oracle, 60, seed=6, read_len=250) > 60
