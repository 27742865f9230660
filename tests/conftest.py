"""pytest configuration: `gpu` marker + shared fixtures (oracle, libraries, simulated data)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return oracle_py.Oracle(so)


@pytest.fixture(scope="session")
def emu_lib():
    """Host-emulation build of the kernel sources (CPU-side logic tests only)."""
    from speedseq_amd import capi
    so = os.path.join(ROOT, "tests", "emu", "libssgpu_emu.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", ROOT, "emu"])
    return capi.Lib(so)


@pytest.fixture(scope="session")
def gpu_lib(gpu_lib_session):
    return gpu_lib_session


@pytest.fixture(scope="session")
def gpu_lib_session():
    from speedseq_amd import capi
    lib = capi.Lib()  # raises if the HIP build is missing: no fallback
    if lib.device_count() < 1:
        pytest.fail("libssgpu.so loaded but no HIP device is visible")
    return lib


@pytest.fixture(scope="session")
def repeat_prefix(oracle, tmp_path_factory):
    import common
    return common.repeat_reference(oracle, tmp_path_factory.mktemp("repeats"))


@pytest.fixture(scope="session")
def repeat_mid_prefix(oracle, tmp_path_factory):
    """Small repeat families (a dozen copies, some of them identical: equal positions on the query, equal chain weights): reads with 10 to 63
    seeds -- the classes of the light reads' LDS chaining kernel."""
    import common
    return common.repeat_reference(oracle, tmp_path_factory.mktemp("repeats_mid"), seed=11, n_copies=(14, 9, 5), fam_len=(500, 320, 260), unique=30000, max_div=0.02)


@pytest.fixture(scope="session")
def repeat_pe_prefix(oracle, tmp_path_factory):
    """Repeat families inside enough unique sequence for the insert-size model to succeed: mate rescue then
    runs against region lists of hundreds of entries (incremental re-sort path)."""
    import common
    return common.repeat_reference(oracle, tmp_path_factory.mktemp("repeats_pe"), unique=300000)
