#!/bin/bash
# Round 4, 23rd GPU call: the bench line of the tree as shipped (the LDS chaining form of r04u is gone: slower; every other kernel's ISA is what r04u's suite ran), kernel pin.
out=$PWD/gpurun_out; mkdir -p $out
python tools/isa_pin.py --check --golden tests/golden/kernel_isa_r04u.sha256 | tail -2
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-script-pairs 0 > $out/r04w_bench.json 2> $out/r04w_bench.err; tail -2 $out/r04w_bench.err
python - <<'PY'
import json, subprocess, sys
d=json.load(open('gpurun_out/r04w_bench.json')); L=d.get('literal',{})
r=d.get('roofline',{})
print('ms/step', round(d['ms_per_step'],1), 'value', d.get('value'), 'literal', d.get('value_literal',{}).get('value'), 'parity', d.get('parity',{}).get('parity_ok'), 'bwt_extends', d['config']['bwt_extends'])
print('roofline', {k:r.get(k) for k in ('kernel','achieved','frac','traffic','ms_per_launch','largest_kernel')})
for k in ('fused','text'):
    x=L.get(k,{}); print(k, {y:x.get(y) for y in ('pairs','wall_s','pairs_per_s','error')})
print('config5', {k:d.get('config5',{}).get(k) for k in ('ms_per_step','pairs_per_s','parity_ok')})
if d.get('parity',{}).get('parity_ok'):
    print(subprocess.run([sys.executable,'tools/isa_pin.py','--write','--golden','gpurun_out/r04w_kernel_isa.sha256'],capture_output=True,text=True).stdout)
PY
