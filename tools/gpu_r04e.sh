#!/bin/bash
# Round 4, fifth GPU call: give-up criteria of the lane kernel (extension budget x longest row after a forward pass) inside the step.
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "smem" > $out/r04e_pytest_smem.log 2>&1; tail -2 $out/r04e_pytest_smem.log
cfg=""
for e in 2000 3000; do for r in 24 32 48 64 96; do cfg="$cfg e${e}r${r}:SSG_SMEM_MAX_EXT=$e,SSG_SMEM_MAX_ROW=$r"; done; done
timeout 900 python tools/smem_ab.py --out $out/r04e_smem_ab.json s2 b2000:SSG_SMEM_MAX_EXT=2000 $cfg e1500r32:SSG_SMEM_MAX_EXT=1500,SSG_SMEM_MAX_ROW=32 e4096r32:SSG_SMEM_MAX_EXT=4096,SSG_SMEM_MAX_ROW=32 e2000r32w12:SSG_SMEM_MAX_EXT=2000,SSG_SMEM_MAX_ROW=32,SSG_SMEM_WAVES_PER_CU=12 > $out/r04e_smem_ab.log 2>&1
grep -E "\"config\"|summary counts|Error|error" $out/r04e_smem_ab.log | cut -c12-250
