#!/bin/bash
# round 2, pass h: ranked (bitmap) chaining: parity tests (each under its own timeout), then the bench
out=$PWD/gpurun_out; mkdir -p $out
for t in test_gpu_repeats_align1 test_gpu_repeats_pe_sam test_gpu_repeats_mate_rescue test_gpu_pair_wave_kernel_forced test_gpu_pe_sam_150 test_gpu_hotpath_batches_and_dups; do
  timeout 120 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k $t 2>&1 | tail -1
done
timeout 240 python bench.py --steps 3 --warmup 1 --no-e2e --cpu-sample 2000 > $out/r02h_var.json 2> $out/r02h_var.err || tail -5 $out/r02h_var.err
python - <<'PY'
import json,sys
d=json.load(open('gpurun_out/r02h_var.json'))
k=d['roofline']['kernels_ms_per_step']
print('ms/step', round(d['ms_per_step'],1), d['value'], d['roofline']['kernel'], d['roofline']['frac'], d.get('parity',{}).get('parity_ok'), {x:k[x] for x in list(k)[:12]})
PY
