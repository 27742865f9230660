#!/bin/bash
# round 2, pass k: leaner SW rows (DPP max folded, score registers, ballot instead of a row maximum), ext_lane two columns per trip
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3
timeout 240 python bench.py --steps 3 --warmup 1 --no-e2e --cpu-sample 2000 > $out/r02k.json 2> $out/r02k.err || tail -5 $out/r02k.err
python - <<'PY'
import json,sys
d=json.load(open('gpurun_out/r02k.json'))
k=d['roofline']['kernels_ms_per_step']
print('ms/step', round(d['ms_per_step'],1), d['value'], d['roofline']['kernel'], d['roofline']['frac'], d.get('parity',{}).get('parity_ok'), {x:k[x] for x in list(k)[:14]})
PY
