#!/bin/bash
# round 2, pass h: ranked (bitmap) chaining: parity tests, then the bench
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $out/r02h_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $out/r02h_pytest.log
for v in "SSG_CHAIN_RANKED=1" "SSG_CHAIN_RANKED=0"; do
  env $v timeout 240 python bench.py --steps 3 --warmup 1 --no-e2e --cpu-sample 0 > $out/r02h_var.json 2> $out/r02h_var.err || tail -5 $out/r02h_var.err
  python - "$v" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r02h_var.json'))
k=d['roofline']['kernels_ms_per_step']
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],1), round(d['value']), {x:k[x] for x in k if 'chain' in x or 'chw' in x})
PY
done
