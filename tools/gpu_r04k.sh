#!/bin/bash
# Round 4, eleventh GPU call: seeding with the table of short-pattern intervals and the third pass first; the device deflate after its regions rewrite.
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_bgzf_device.py tests/test_sambamba.py -m gpu -x -q -k "smem or seeds or align1_150 or bgzf or sambamba or repeats_align1 or extend_lane" > $out/r04k_pytest.log 2>&1; tail -3 $out/r04k_pytest.log
timeout 300 python tools/dbg/bgzf_bench.py 1024 2>&1 | tee $out/r04k_bgzf_bench.log | tail -5
timeout 900 python tools/smem_ab.py --out $out/r04k_smem_ab.json kt nokt:SSG_SMEM_USE_KTAB=0 kt_w16:SSG_SMEM_WAVES_PER_CU=16 kt_w8:SSG_SMEM_WAVES_PER_CU=8 kt_e3000r48:SSG_SMEM_MAX_EXT=3000,SSG_SMEM_MAX_ROW=48 kt_e1500r24:SSG_SMEM_MAX_EXT=1500,SSG_SMEM_MAX_ROW=24 kt_e2000r24:SSG_SMEM_MAX_ROW=24 kt_e2000r48:SSG_SMEM_MAX_ROW=48 \
  kt_h32:SSG_SMEM_HEAVY_WAVES_PER_CU=32 h8@seed_h8:SSG_SMEM_HEAVY_WAVES_PER_CU=32 kt_tune:SSG_S2_TUNE=1 > $out/r04k_smem_ab.log 2>&1
grep -E "\"config\"|summary counts|Error|error|timeline|table of" $out/r04k_smem_ab.log | cut -c12-330
