"""tools/fuzz_pipeline.py as tests: the text hand-off against the fused one with every host-side switch drawn at random -- a few seeds on the emulation build here,
a few on the product's executables on the MI355X (several emulated-device settings have no meaning there: the switches that remain are calls in flight, formatter
threads, frame segments, reader threads, densification, sort spills, samblaster's options)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_fuzz_pipeline_emulated(emu_lib):
    import fuzz_pipeline
    assert fuzz_pipeline.main(runs=2, first_seed=20) == 0


@pytest.mark.gpu
def test_fuzz_pipeline_gpu(gpu_lib):
    import fuzz_pipeline
    assert fuzz_pipeline.main(runs=4, bindir=os.path.join(ROOT, "bin"), first_seed=30) == 0
