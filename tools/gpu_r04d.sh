#!/bin/bash
# Round 4, fourth GPU call: the wave-per-read kernel for the reads the lane kernel gives up: parity tests, then time inside the step by budget.
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "smem or align1_150 or repeats_align1" > $out/r04d_pytest_smem.log 2>&1; tail -3 $out/r04d_pytest_smem.log
timeout 900 python tools/smem_ab.py --out $out/r04d_smem_ab.json s2 b1000:SSG_SMEM_MAX_EXT=1000 b1500:SSG_SMEM_MAX_EXT=1500 b2000:SSG_SMEM_MAX_EXT=2000 b3000:SSG_SMEM_MAX_EXT=3000 b4096:SSG_SMEM_MAX_EXT=4096 \
  b2000h8:SSG_SMEM_MAX_EXT=2000,SSG_SMEM_HEAVY_WAVES_PER_CU=8 b2000h32:SSG_SMEM_MAX_EXT=2000,SSG_SMEM_HEAVY_WAVES_PER_CU=32 b2000w12:SSG_SMEM_MAX_EXT=2000,SSG_SMEM_WAVES_PER_CU=12 b1500w12:SSG_SMEM_MAX_EXT=1500,SSG_SMEM_WAVES_PER_CU=12 > $out/r04d_smem_ab.log 2>&1
grep -E "\"config\"|summary counts|Error|error" $out/r04d_smem_ab.log | cut -c1-330
