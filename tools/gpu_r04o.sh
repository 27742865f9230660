#!/bin/bash
# Round 4, 15th GPU call: instruction-side variants of the lane seeding kernel (state-machine trips, a ballot before the second trip, five waves per SIMD).
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python tools/smem_ab.py --out $out/r04o_smem_ab.json base tb@seed_tb t1@seed_t1 t3@seed_t3 w5@seed_w5:SSG_SMEM_WAVES_PER_CU=20 base2 > $out/r04o_smem_ab.log 2>&1
grep -E "\"config\"|summary counts|Error|error" $out/r04o_smem_ab.log | cut -c12-330
