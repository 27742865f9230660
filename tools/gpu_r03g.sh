#!/bin/bash
# Round 3, fifth GPU call: the seeding kernel back on its round-2 code path by default (the table path is a separate template instance), GPU suite,
# densify self-check at 3.1 Gbp, bench legs at reduced sizes, then the table path (K = 13) on the bench step with its parity gate.
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q > $out/r03g_pytest_gpu.log 2>&1; tail -2 $out/r03g_pytest_gpu.log
SSG_KTAB_K=7 timeout 300 python -m pytest tests -m gpu -q -k "smem or cli_gpu or align1" > $out/r03g_pytest_ktab.log 2>&1; tail -3 $out/r03g_pytest_ktab.log
echo skip-densify-check
timeout 400 python bench.py --steps 3 --warmup 1 --cpu-sample 20000 --e2e-pairs 4000000 --script-pairs 4000000 --cpu-script-pairs 50000 --partial $out/r03g_partial.json > $out/r03g_bench.json 2> $out/r03g_bench.err; echo "bench rc=$?"
grep "^\[bench" $out/r03g_bench.err | tail -30
python - <<'PY'
import json,os
p='gpurun_out/r03g_bench.json'
d=json.load(open(p if os.path.getsize(p) else 'gpurun_out/r03g_partial.json'))
print('value', d['value'], 'ms/step', round(d['ms_per_step'],1), 'parity', json.dumps(d.get('parity',{}))[:400])
k=d.get('roofline',{}).get('kernels_ms_per_step',{}); print({x:k[x] for x in list(k)[:10]}, 'frac', d.get('roofline',{}).get('frac'))
print('cpu', json.dumps(d.get('cpu_baseline',{}))[:900])
e=d.get('e2e',{}); print('e2e', {k:e.get(k) for k in ('index_load_s','reads_to_sam_s','pairs_per_s','bwa_stage_busy','pairs_per_s_gz_input','sample_streams_identical','error')})
print('literal', json.dumps(d.get('literal',{}),indent=1)[:3500])
PY
SSG_KTAB_K=13 timeout 200 python bench.py --steps 3 --warmup 1 --cpu-sample 20000 --no-e2e --partial $out/r03g_k13.json > $out/r03g_k13_bench.json 2> $out/r03g_k13.err; echo "k13 rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03g_k13_bench.json'))
k=d.get('roofline',{}).get('kernels_ms_per_step',{})
print('K=13: value', d['value'], 'ms/step', round(d['ms_per_step'],1), 'parity_ok', d.get('parity',{}).get('parity_ok'), {x:k[x] for x in list(k)[:6]})
PY
