#!/bin/bash
# round 2, pass q: SMEM state machine: hot states first in the dispatch, FWDEND / RET / P3 transitions done where they arise
out=$PWD/gpurun_out; mkdir -p $out
for t in test_gpu_smem test_gpu_repeats_align1 test_gpu_pe_sam_150; do
  timeout 120 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k $t 2>&1 | tail -1
done
timeout 240 python bench.py --steps 3 --warmup 1 --no-e2e --cpu-sample 2000 > $out/r02q.json 2> $out/r02q.err || tail -5 $out/r02q.err
python - <<'PY'
import json,sys
d=json.load(open('gpurun_out/r02q.json'))
k=d['roofline']['kernels_ms_per_step']
print('ms/step', round(d['ms_per_step'],1), d.get('parity',{}).get('parity_ok'), {x:k[x] for x in list(k)[:6]})
PY
