"""The drop-in boundary: speedseq_amd/libssgpu.so must load and export every entry point include/ssgpu.h declares (so must the
host-emulation build the CPU-side tests use), and without a HIP device the product library must refuse to compute -- there is no
CPU fallback behind the C ABI."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from common import EXAMPLE_FA, ROOT

HDR = os.path.join(ROOT, "include", "ssgpu.h")


def declared():
    txt = re.sub(r"/\*.*?\*/", " ", open(HDR).read(), flags=re.S)
    txt = "\n".join(l for l in txt.split("\n") if not l.lstrip().startswith("#"))
    names = set(re.findall(r"\b(ssg_[a-z0-9_]+)\s*\(", txt))
    assert len(names) > 40
    return sorted(names)


@pytest.mark.parametrize("path", ["speedseq_amd/libssgpu.so", "tests/emu/libssgpu_emu.so"])
def test_library_exports_every_declared_entry_point(path, emu_lib):
    lib = C.CDLL(os.path.join(ROOT, path))
    missing = [n for n in declared() if not hasattr(lib, n)]
    assert not missing, missing
    lib.ssg_backend.restype = C.c_char_p
    assert lib.ssg_backend() in (b"hip:gfx950", b"emu")


def test_product_library_has_no_cpu_path():
    lib = C.CDLL(os.path.join(ROOT, "speedseq_amd", "libssgpu.so"))
    lib.ssg_backend.restype = C.c_char_p
    lib.ssg_last_error.restype = C.c_char_p
    assert lib.ssg_backend() == b"hip:gfx950"
    if lib.ssg_device_count() > 0:
        pytest.skip("a HIP device is visible: the refusal path is for machines without one")
    idx = C.c_void_p()
    rc = lib.ssg_index_load(EXAMPLE_FA.encode(), C.byref(idx))
    assert rc != 0 and b"no HIP device" in lib.ssg_last_error()
    keys = np.arange(8, dtype=np.uint64)
    perm = np.zeros(8, dtype=np.uint32)
    assert lib.ssg_sort_u64_perm(keys.ctypes.data_as(C.c_void_p), C.c_int64(8), perm.ctypes.data_as(C.c_void_p)) != 0
    assert lib.ssg_pe_reserve(1000, 1) != 0
