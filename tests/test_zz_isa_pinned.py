"""The product kernels' gfx950 machine code is the code whose results were checked on the MI355X (tools/isa_pin.py): an alarm for kernels that
changed -- in source, or only in the code the compiler made of them -- since the last GPU run, which the CPU-side suite cannot judge."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_kernels_machine_code_is_the_gpu_checked_build():
    lib = os.path.join(ROOT, "speedseq_amd", "libssgpu.so")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") or not os.path.exists(lib):
        pytest.skip("llvm-objdump or libssgpu.so not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_pin.py")], capture_output=True, text=True)
    assert r.returncode == 0, "kernels whose machine code has not been through `pytest -m gpu` + the parity gate yet (run a GPU script; it writes the new pin):\n" + r.stdout[-3000:]
