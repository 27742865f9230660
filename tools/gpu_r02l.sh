#!/bin/bash
# round 2, pass l: chain class 512, mate-rescue target bases by readlane; A/B: SMEM at 5 waves/SIMD, mate rescue at 4 waves/SIMD
out=$PWD/gpurun_out; mkdir -p $out
for t in test_gpu_repeats_align1 test_gpu_repeats_mate_rescue test_gpu_pe_sam_150; do
  timeout 120 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k $t 2>&1 | tail -1
done
run() { # name, env...
  n=$1; shift
  env "$@" timeout 240 python bench.py --steps 3 --warmup 1 --no-e2e --cpu-sample $CS > $out/r02l_$n.json 2> $out/r02l_$n.err || tail -5 $out/r02l_$n.err
  python - $n <<'PY'
import json,sys
d=json.load(open('gpurun_out/r02l_%s.json' % sys.argv[1]))
k=d['roofline']['kernels_ms_per_step']
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],1), d.get('parity',{}).get('parity_ok'), {x:k[x] for x in list(k)[:16]})
PY
}
CS=2000 run default X=1
CS=0 run smq5 SSGPU_LIB=$PWD/speedseq_amd/libssgpu_smq5.so SSG_SMEM_WAVES_PER_CU=20
CS=0 run sw4 SSGPU_LIB=$PWD/speedseq_amd/libssgpu_sw4.so
