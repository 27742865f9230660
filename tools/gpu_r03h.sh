#!/bin/bash
# Round 3, GPU call after the bisect: main translation unit back to the machine code of the last good run (checked kernel by kernel), the
# optional table in a unit of its own.  Sanity first (stop if the step is still wrong), then the full bench line, then the table instance.
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q > $out/r03h_pytest_gpu.log 2>&1; tail -2 $out/r03h_pytest_gpu.log
timeout 200 python bench.py --steps 2 --warmup 1 --cpu-sample 20000 --no-e2e --partial $out/r03h_sanity.json > $out/r03h_sanity_bench.json 2> $out/r03h_sanity.err
python - <<'PY' || exit 0
import json,sys
d=json.load(open('gpurun_out/r03h_sanity_bench.json'))
ok=d.get('parity',{}).get('parity_ok')
print('sanity: ms/step', round(d['ms_per_step'],1), 'parity_ok', ok)
sys.exit(0 if ok and d['ms_per_step']<600 else 1)
PY
SSG_KTAB_K=13 timeout 200 python bench.py --steps 3 --warmup 1 --cpu-sample 20000 --no-e2e --partial $out/r03h_k13.json > $out/r03h_k13_bench.json 2> $out/r03h_k13.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03h_k13_bench.json'))
k=d.get('roofline',{}).get('kernels_ms_per_step',{})
print('K=13: value', d['value'], 'ms/step', round(d['ms_per_step'],1), 'parity_ok', d.get('parity',{}).get('parity_ok'), {x:k[x] for x in list(k)[:6]})
PY
timeout 700 python bench.py --steps 10 --warmup 3 --partial $out/r03h_partial.json > $out/r03h_bench.json 2> $out/r03h_bench.err; echo "bench rc=$?"
grep "^\[bench" $out/r03h_bench.err | tail -30
python - <<'PY'
import json,os
p='gpurun_out/r03h_bench.json'
d=json.load(open(p if os.path.getsize(p) else 'gpurun_out/r03h_partial.json'))
print('value', d['value'], 'ms/step', round(d['ms_per_step'],1), 'parity', json.dumps(d.get('parity',{}))[:500])
k=d.get('roofline',{}).get('kernels_ms_per_step',{}); print({x:k[x] for x in list(k)[:10]}, 'frac', d.get('roofline',{}).get('frac'))
print('cpu', json.dumps(d.get('cpu_baseline',{}))[:1200])
e=d.get('e2e',{}); print('e2e', {k:e.get(k) for k in ('index_load_s','reads_to_sam_s','pairs_per_s','bwa_stage_busy','pairs_per_s_gz_input','sample_streams_identical','error')})
print('literal', json.dumps(d.get('literal',{}),indent=1)[:4000])
PY
