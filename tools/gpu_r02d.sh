#!/bin/bash
# round 2, pass d: cooperative-fetch SMEM kernel: parity tests, then A/B at the headline size
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $out/r02d_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/r02d_pytest.log
for v in "SSG_SMEM_COOP=0" "SSG_SMEM_WAVES_PER_CU=20" "SSG_SMEM_WAVES_PER_CU=12"; do
  env $v timeout 600 python bench.py --steps 2 --warmup 1 --no-e2e --cpu-sample 0 > $out/r02d_var.json 2> $out/r02d_var.err || tail -5 $out/r02d_var.err
  python - "$v" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r02d_var.json'))
k=d['roofline']['kernels_ms_per_step']
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],1), {x:k[x] for x in k if 'smem' in x}, d['roofline']['random64B'])
PY
done
