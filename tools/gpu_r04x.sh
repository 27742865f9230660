#!/bin/bash
# Round 4, 24th GPU call (last minutes of the budget): the host-side changes made after the last suite run -- samblaster's two stages, the sort's
# producers on every device -- through the tests that run those executables on the GPU.
out=$PWD/gpurun_out; mkdir -p $out
timeout 150 python -m pytest tests/test_fused.py tests/test_sambamba.py tests/test_speedseq_script.py -m gpu -x -q > $out/r04x_pytest_host_changes.log 2>&1; tail -3 $out/r04x_pytest_host_changes.log
