#!/bin/bash
# Round 4, third GPU call: the seeding kernel's time inside the step with a per-read extension budget (product instance, no counters): how much
# of the launch is the tail behind the heaviest reads.  The given-up reads go through the nested-loop lane kernel here (slow; its time is listed apart).
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python tools/smem_ab.py --out $out/r04c_smem_ab.json s2 b8192:SSG_SMEM_MAX_EXT=8192 b4096:SSG_SMEM_MAX_EXT=4096 b3000:SSG_SMEM_MAX_EXT=3000 b2000:SSG_SMEM_MAX_EXT=2000 b1500:SSG_SMEM_MAX_EXT=1500 \
  b4096w8:SSG_SMEM_MAX_EXT=4096,SSG_SMEM_WAVES_PER_CU=8 b2000w8:SSG_SMEM_MAX_EXT=2000,SSG_SMEM_WAVES_PER_CU=8 b2000w12:SSG_SMEM_MAX_EXT=2000,SSG_SMEM_WAVES_PER_CU=12 b2000w2@seed_w2:SSG_SMEM_MAX_EXT=2000 > $out/r04c_smem_ab.log 2>&1
grep -E "\"config\"|summary counts|Error|error" $out/r04c_smem_ab.log | cut -c1-330
