#!/bin/bash
# Round 4, 21st GPU call: the whole tree after the command-line work (option letters, -M -Y -S -P, single-end input) and the LDS form of the light reads'
# chaining: suite, chain A/B inside the step, the bench line, the kernel pin if everything is green.
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/r04u_pytest_gpu.log 2>&1; tail -3 $out/r04u_pytest_gpu.log
timeout 300 python tools/smem_ab.py --kernels chain --out $out/r04u_chain_ab.json lds:SSG_CHAIN_LDS=1 global:SSG_CHAIN_LDS=0 lds_again:SSG_CHAIN_LDS=1 > $out/r04u_chain_ab.log 2>&1
grep -E "\"config\"|summary counts" $out/r04u_chain_ab.log | cut -c12-420
timeout 900 python bench.py --steps 5 --warmup 2 --cpu-script-pairs 0 > $out/r04u_bench.json 2> $out/r04u_bench.err; tail -2 $out/r04u_bench.err
python - <<'PY'
import json, subprocess, sys
d=json.load(open('gpurun_out/r04u_bench.json')); L=d.get('literal',{})
r=d.get('roofline',{})
print('ms/step', round(d['ms_per_step'],1), 'value', d.get('value'), 'literal', d.get('value_literal',{}).get('value'), 'parity', d.get('parity',{}).get('parity_ok'), 'bwt_extends', d['config']['bwt_extends'])
print('roofline', {k:r.get(k) for k in ('kernel','achieved','frac','traffic','ms_per_launch','largest_kernel')})
print('kernels', list(r.get('kernels_ms_per_step',{}).items())[:14])
for k in ('fused','text'):
    x=L.get(k,{}); print(k, {y:x.get(y) for y in ('pairs','wall_s','pairs_per_s','error','bam_bytes')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('cores'))
print('config5', {k:d.get('config5',{}).get(k) for k in ('ms_per_step','pairs_per_s','parity_ok')})
log=open('gpurun_out/r04u_pytest_gpu.log').read()
if d.get('parity',{}).get('parity_ok') and ' passed' in log and 'failed' not in log and 'error' not in log.lower():
    print(subprocess.run([sys.executable,'tools/isa_pin.py','--write','--golden','gpurun_out/r04u_kernel_isa.sha256'],capture_output=True,text=True).stdout)
PY
