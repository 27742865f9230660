#!/bin/bash
# Round 4, ninth GPU call: the literal leg after the sort's hand-over was rewritten (atomics instead of one condition variable) with two device calls in
# flight per GPU by default; then the configs[2] soak: 40 M pairs through the reference's script on one GPU.
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 2000 --config5-pairs 0 --cpu-script-pairs 0 --no-dist-rehearsal --no-profile > $out/r04i_bench_literal.json 2> $out/r04i_bench_literal.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04i_bench_literal.json')); L=d.get('literal',{})
print('sample BAMs equal oracle:', L.get('sample_bams_equal_oracle'))
for k in ('fused','text'):
    x=L.get(k,{}); print(k, {y:x.get(y) for y in ('pairs','wall_s','pairs_per_s','error')})
    for l in x.get('stage_log',[]): print('   ', l[:380])
PY
timeout 1500 python tools/soak.py --pairs 40000000 > $out/r04i_soak_40M.json 2> $out/r04i_soak_40M.err; tail -3 $out/r04i_soak_40M.err; head -c 3000 $out/r04i_soak_40M.json
