#!/bin/bash
# round 2, pass c: e2e leg with overlapped stages; SMEM launch-shape variants at the headline size
out=$PWD/gpurun_out; mkdir -p $out
timeout 1200 python bench.py --steps 3 --warmup 1 > $out/r02c_bench.json 2> $out/r02c_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02c_bench.json'))
print(d['value'], d['ms_per_step']); print(json.dumps(d.get('e2e'), indent=1)); print(d.get('parity',{}).get('parity_ok'))
PY
grep -v "ssg index" $out/r02c_bench.err | tail -5
for v in "SSG_SMEM_LPR=1" "SSG_SMEM_WAVES_PER_CU=24" "SSG_SMEM_WAVES_PER_CU=32" "SSG_SMEM_WAVES_PER_CU=12"; do
  env $v timeout 600 python bench.py --steps 2 --warmup 1 --no-e2e --cpu-sample 0 > $out/r02c_var.json 2> $out/r02c_var.err
  python - "$v" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r02c_var.json'))
k=d['roofline']['kernels_ms_per_step']
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],1), {x:k[x] for x in k if 'smem' in x})
PY
done
