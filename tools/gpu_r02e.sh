#!/bin/bash
# round 2, pass e: packed 16-bit striped local SW (mate rescue): parity tests, A/B at the headline size, validators test
out=$PWD/gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_validators.py -m gpu -x -q > $out/r02e_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/r02e_pytest.log
for v in "SSG_SW_INT32=0" "SSG_SW_INT32=1"; do
  env $v timeout 600 python bench.py --steps 2 --warmup 1 --no-e2e --cpu-sample 0 > $out/r02e_var.json 2> $out/r02e_var.err || tail -5 $out/r02e_var.err
  python - "$v" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r02e_var.json'))
k=d['roofline']['kernels_ms_per_step']
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],1), {x:k[x] for x in list(k)[:8]})
PY
done
