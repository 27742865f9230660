#!/bin/bash
# Round 4, eighth GPU call: the tree after the old seeding kernels were removed: suite, bench line as the driver runs it (the sort's write stage with its
# new timers), then the kernel pin of this build if both are green; the literal leg with two device calls in flight per GPU.
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/r04h_pytest_gpu.log 2>&1; tail -3 $out/r04h_pytest_gpu.log
timeout 900 python bench.py --steps 5 --warmup 2 > $out/r04h_bench.json 2> $out/r04h_bench.err; tail -2 $out/r04h_bench.err
python - <<'PY'
import json, subprocess, sys
d=json.load(open('gpurun_out/r04h_bench.json')); L=d.get('literal',{})
r=d.get('roofline',{})
print('ms/step', round(d['ms_per_step'],1), 'value', d.get('value'), 'parity', d.get('parity',{}).get('parity_ok'), 'roofline', {k:r.get(k) for k in ('kernel','achieved','frac','traffic','ms_per_launch','largest_kernel')})
for k in ('fused','text'):
    x=L.get(k,{}); print(k, {y:x.get(y) for y in ('pairs','wall_s','pairs_per_s','error')})
    for l in x.get('stage_log',[]): print('   ', l[:420])
log=open('gpurun_out/r04h_pytest_gpu.log').read()
if d.get('parity',{}).get('parity_ok') and ' passed' in log and 'failed' not in log:
    print(subprocess.run([sys.executable,'tools/isa_pin.py','--write','--golden','gpurun_out/r04h_kernel_isa.sha256'],capture_output=True,text=True).stdout)
PY
SSG_BENCH_CONFIG_EXTRA="export SSG_BWA_INFLIGHT=2" timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 2000 --config5-pairs 0 --cpu-script-pairs 0 --no-dist-rehearsal --no-profile > $out/r04h_bench_inflight2.json 2> $out/r04h_bench_inflight2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04h_bench_inflight2.json')); r=d.get('literal',{}).get('fused',{})
print('two calls in flight per GPU:', {x:r.get(x) for x in ('pairs','wall_s','pairs_per_s','error')}, 'sample BAMs equal oracle:', d.get('literal',{}).get('sample_bams_equal_oracle'))
for l in r.get('stage_log',[])[:5]: print('   ', l[:300])
PY
