#!/bin/bash
# round 2, pass n: SMEM state machine with a bounded number of steps per extension round (1 and 2)
out=$PWD/gpurun_out; mkdir -p $out
for t in test_gpu_smem test_gpu_repeats_align1; do
  timeout 120 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k $t 2>&1 | tail -1
done
run() { n=$1; shift
  env "$@" timeout 240 python bench.py --steps 3 --warmup 1 --no-e2e --cpu-sample $CS > $out/r02n_$n.json 2> $out/r02n_$n.err || tail -5 $out/r02n_$n.err
  python - $n <<'PY'
import json,sys
d=json.load(open('gpurun_out/r02n_%s.json' % sys.argv[1]))
k=d['roofline']['kernels_ms_per_step']
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],1), d.get('parity',{}).get('parity_ok'), {x:k[x] for x in list(k)[:6]})
PY
}
CS=2000 run trips1 X=1
CS=0 run trips2 SSGPU_LIB=$PWD/speedseq_amd/libssgpu_trips2.so
