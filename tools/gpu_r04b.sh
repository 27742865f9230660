#!/bin/bash
# Round 4, second GPU call: (1) which part of the "uniform" control flow the seeding kernel needs to be right on the MI355X; (2) the new
# seeding kernel (ssg_seed.cpp, k_smem2.h) as the default: GPU suite; (3) its time inside the step against the round-3 kernel, the launch
# timeline (when the pool of reads runs dry, when the last lane is done), reads by extension count, give-up budgets, compile-time variants.
out=$PWD/gpurun_out; mkdir -p $out
KT_KS="8" bash tools/dbg/kt_variants.sh run > /dev/null 2>&1; mv $out/kt_variants.log $out/r04b_kt_variants.log; grep -E "^==|differ;|counters" $out/r04b_kt_variants.log
timeout 900 python -m pytest tests -m gpu -x -q > $out/r04b_pytest_gpu.log 2>&1; tail -3 $out/r04b_pytest_gpu.log
timeout 900 python tools/smem_ab.py --out $out/r04b_smem_ab.json quad:SSG_SMEM_KERNEL=quad s2 s2tune:SSG_S2_TUNE=1 b6000:SSG_SMEM_MAX_EXT=6000,SSG_S2_TUNE=1 b3000:SSG_SMEM_MAX_EXT=3000,SSG_S2_TUNE=1 b2000:SSG_SMEM_MAX_EXT=2000,SSG_S2_TUNE=1 \
  wcu12:SSG_SMEM_WAVES_PER_CU=12 wcu8:SSG_SMEM_WAVES_PER_CU=8 w2@seed_w2 w3@seed_w3 t1@seed_t1 t3@seed_t3 > $out/r04b_smem_ab.log 2>&1
grep -E "smem2 timeline|smem2 reads|\"config\"|summary counts|Error|error" $out/r04b_smem_ab.log | cut -c1-420
