#!/bin/bash
# Round 3, second GPU call: the fused plugin path + streamed index load + LF-walk SA densify on the GPU; full bench line with the literal metric.
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/r03b_pytest_gpu.log 2>&1; tail -4 $out/r03b_pytest_gpu.log
timeout 1500 python bench.py --steps 5 --warmup 2 > $out/r03b_bench.json 2> $out/r03b_bench.err; echo "bench rc=$?"; tail -5 $out/r03b_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03b_bench.json'))
print('value', d['value'], 'ms/step', round(d['ms_per_step'],1), 'parity', json.dumps(d.get('parity',{}))[:900])
print('cpu', json.dumps(d.get('cpu_baseline',{}))[:700])
e=d.get('e2e',{}); print('e2e', {k:e.get(k) for k in ('index_load_s','reads_to_sam_s','pairs_per_s','bwa_stage_busy','pairs_per_s_gz_input','sample_streams_identical','error')})
print('literal', json.dumps(d.get('literal',{}),indent=1)[:3500])
PY
