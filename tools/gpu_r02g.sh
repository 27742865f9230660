#!/bin/bash
# round 2, pass g: concurrent parts in the hot path: parity, then A/B
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $out/r02g_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $out/r02g_pytest.log
for v in "SSG_HOTPATH_PARTS=1" "SSG_HOTPATH_PARTS=2" "SSG_HOTPATH_PARTS=4"; do
  env $v timeout 600 python bench.py --steps 3 --warmup 1 --no-e2e --cpu-sample 0 --bwa-threads 8 > $out/r02g_var.json 2> $out/r02g_var.err || tail -5 $out/r02g_var.err
  python - "$v bwa-threads 8 (4 upstream batches)" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r02g_var.json'))
k=d['roofline']['kernels_ms_per_step']
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],1), round(d['value']), {x:k[x] for x in list(k)[:6]})
PY
done
for v in "SSG_HOTPATH_PARTS=1" "SSG_HOTPATH_PARTS=2"; do
  env $v timeout 600 python bench.py --steps 3 --warmup 1 --no-e2e --cpu-sample 0 > $out/r02g_var.json 2> $out/r02g_var.err || tail -5 $out/r02g_var.err
  python - "$v default -t 16 (2 upstream batches)" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r02g_var.json'))
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],1), round(d['value']))
PY
done
