#!/bin/bash
# Round 3, third GPU call: short legs with progress logs (where does the bench spend its time?), the k-mer interval table and the pass-3 skip in the SMEM kernel, lane-busy counters.
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q > $out/r03c_pytest_gpu.log 2>&1; tail -3 $out/r03c_pytest_gpu.log
timeout 560 python bench.py --steps 3 --warmup 1 --cpu-sample 50000 --e2e-pairs 4000000 --script-pairs 4000000 --cpu-script-pairs 50000 --partial $out/r03c_partial.json > $out/r03c_bench.json 2> $out/r03c_bench.err; echo "bench rc=$?"
grep "^\[bench" $out/r03c_bench.err | tail -30
python - <<'PY'
import json,os
p='gpurun_out/r03c_bench.json'
d=json.load(open(p if os.path.getsize(p) else 'gpurun_out/r03c_partial.json'))
print('value', d['value'], 'ms/step', round(d['ms_per_step'],1), 'parity', json.dumps(d.get('parity',{}))[:600])
k=d.get('roofline',{}).get('kernels_ms_per_step',{}); print({x:k[x] for x in list(k)[:10]}, 'frac', d.get('roofline',{}).get('frac'))
print('cpu', json.dumps(d.get('cpu_baseline',{}))[:900])
e=d.get('e2e',{}); print('e2e', {k:e.get(k) for k in ('index_load_s','reads_to_sam_s','pairs_per_s','bwa_stage_busy','pairs_per_s_gz_input','sample_streams_identical','error')})
print('literal', json.dumps(d.get('literal',{}),indent=1)[:3000])
PY
timeout 240 python tools/dbg/phase.py > $out/r03c_phase.txt 2>&1; tail -3 $out/r03c_phase.txt
