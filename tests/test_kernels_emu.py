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


def test_emu_align1_250(emu_lib, oracle):
    assert common.check_align1(emu_lib, oracle, 60, seed=6, read_len=250) > 60
