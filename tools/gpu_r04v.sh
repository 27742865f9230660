#!/bin/bash
# Round 4, 22nd GPU call: where the light reads' chaining time really is.  r04u: the LDS form of the reads with <= 12 seeds costs 20 ms and leaves ssg_k_chain at its
# 25 ms -- the reads of 13..63 seeds are its whole time.  Those can go to the wave-per-read kernels (state in LDS) by an existing switch: SSG_CHAIN_WAVE_MIN.
out=$PWD/gpurun_out; mkdir -p $out
timeout 400 python tools/smem_ab.py --kernels chain --out $out/r04v_chain_ab.json base:SSG_CHAIN_LDS=0 w16:SSG_CHAIN_LDS=0,SSG_CHAIN_WAVE_MIN=16 w13lds:SSG_CHAIN_LDS=1,SSG_CHAIN_WAVE_MIN=13 w24:SSG_CHAIN_LDS=0,SSG_CHAIN_WAVE_MIN=24 w32:SSG_CHAIN_LDS=0,SSG_CHAIN_WAVE_MIN=32 w8:SSG_CHAIN_LDS=0,SSG_CHAIN_WAVE_MIN=8 > $out/r04v_chain_ab.log 2>&1
grep -E "\"config\"|summary counts" $out/r04v_chain_ab.log | cut -c12-470
