#!/bin/bash
# Round 4, 26th GPU call (what is left of the budget): rank mode on the MI355X -- two pipelines side by side on the one GPU against one pipeline.
out=$PWD/gpurun_out; mkdir -p $out
timeout 80 python -m pytest tests/test_ranks.py -m gpu -x -q > $out/r04z_pytest_ranks_gpu.log 2>&1; tail -15 $out/r04z_pytest_ranks_gpu.log | cut -c1-300
