#!/bin/bash
# Round 4, sixth GPU call: waves of the lane kernel that run out of reads take given-up reads while the launch finishes (tail phase) vs the kernel behind it alone.
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "smem" > $out/r04f_pytest_smem.log 2>&1; tail -2 $out/r04f_pytest_smem.log
cfg=""
for e in 1500 2000 3000; do for r in 32 48; do cfg="$cfg e${e}r${r}:SSG_SMEM_MAX_EXT=$e,SSG_SMEM_MAX_ROW=$r e${e}r${r}w12:SSG_SMEM_MAX_EXT=$e,SSG_SMEM_MAX_ROW=$r,SSG_SMEM_WAVES_PER_CU=12"; done; done
timeout 900 python tools/smem_ab.py --out $out/r04f_smem_ab.json s2 notail:SSG_SMEM_MAX_EXT=2000,SSG_SMEM_MAX_ROW=32,SSG_SMEM_TAIL_PHASE=0 $cfg e1000r24w12:SSG_SMEM_MAX_EXT=1000,SSG_SMEM_MAX_ROW=24,SSG_SMEM_WAVES_PER_CU=12 e2000r32w8:SSG_SMEM_MAX_EXT=2000,SSG_SMEM_MAX_ROW=32,SSG_SMEM_WAVES_PER_CU=8 tune:SSG_SMEM_MAX_EXT=2000,SSG_SMEM_MAX_ROW=32,SSG_S2_TUNE=1 > $out/r04f_smem_ab.log 2>&1
grep -E "\"config\"|summary counts|Error|error|timeline" $out/r04f_smem_ab.log | cut -c12-250
