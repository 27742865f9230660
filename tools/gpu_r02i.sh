#!/bin/bash
# round 2, pass i: SMEM extension with one memory round trip and same-block reuse: parity tests, then the bench (16 and 20 waves per CU)
out=$PWD/gpurun_out; mkdir -p $out
for t in test_gpu_smem test_gpu_repeats_align1 test_gpu_pe_sam_150; do
  timeout 120 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k $t 2>&1 | tail -1
done
for w in 16 20; do
SSG_SMEM_WAVES_PER_CU=$w timeout 240 python bench.py --steps 3 --warmup 1 --no-e2e --cpu-sample 2000 > $out/r02i_w$w.json 2> $out/r02i_w$w.err || tail -5 $out/r02i_w$w.err
python - $w <<'PY'
import json,sys
d=json.load(open('gpurun_out/r02i_w%s.json' % sys.argv[1]))
k=d['roofline']['kernels_ms_per_step']
print('waves', sys.argv[1], 'ms/step', round(d['ms_per_step'],1), d['value'], d['roofline']['kernel'], d['roofline']['frac'], d.get('parity',{}).get('parity_ok'), {x:k[x] for x in list(k)[:12]})
PY
done
