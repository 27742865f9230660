"""`bwa mem | samblaster` on the MI355X over many upstream batches and device calls (the CPU-side twin is
tests/test_cli.py::test_cli_emu_many_batches_and_device_calls).  Collected last: it was added after the round's last GPU run."""
import os

import pytest

from common import ROOT
import test_cli


@pytest.mark.gpu
def test_cli_gpu_many_batches_and_device_calls(tmp_path, gpu_lib, monkeypatch):
    monkeypatch.setenv("SSG_BWA_CHUNK_BASES", "20000")
    monkeypatch.setenv("ORC_CHUNK_BASES", "20000")
    monkeypatch.setenv("SSG_BWA_CALL_PAIRS", "700")
    monkeypatch.setenv("SSG_SBL_CHUNK", "313")
    b = os.path.join(ROOT, "bin")
    test_cli._check([os.path.join(b, "bwa")], [os.path.join(b, "samblaster")], tmp_path, 5000, seed=30)
