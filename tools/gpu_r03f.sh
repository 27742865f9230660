#!/bin/bash
# Round 3, sixth GPU call: which change broke the seeding results on the GPU?  The same two checks against libraries built from five commits.
out=$PWD/gpurun_out; mkdir -p $out
for v in speedseq_amd/libssgpu_v0_c6012b6.so speedseq_amd/libssgpu_v_6927481.so speedseq_amd/libssgpu_v_9b56c47.so speedseq_amd/libssgpu_v_5711774.so speedseq_amd/libssgpu.so; do
  echo "=== $v"
  SSGPU_LIB=$PWD/$v SSG_KTAB_K=0 timeout 200 python -m pytest tests/test_gpu_kernels.py -q -k "test_gpu_smem or test_gpu_align1_150 or test_gpu_local or test_gpu_extend" 2>&1 | tail -3
  SSGPU_LIB=$PWD/$v SSG_KTAB_K=0 timeout 200 python bench.py --steps 2 --warmup 1 --no-e2e --cpu-sample 0 --no-profile --partial $out/r03f_p.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench ms/step', round(d['ms_per_step'],1), 'records', d['config']['records'], 'seeds', d['config']['seeds'], 'bwt_extends', d['config']['bwt_extends'])"
done
