#!/bin/bash
# Round 4, seventh GPU call: the tree with the new seeding stage as default: GPU suite, the bench line as the driver runs it, kernel-trace stats.
out=$PWD/gpurun_out; mkdir -p $out; repo=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > $out/r04g_pytest_gpu.log 2>&1; tail -3 $out/r04g_pytest_gpu.log
timeout 900 python bench.py --steps 5 --warmup 2 > $out/r04g_bench.json 2> $out/r04g_bench.err; tail -3 $out/r04g_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04g_bench.json')); L=d.get('literal',{})
print('ms/step', round(d['ms_per_step'],1), 'value', d.get('value'), 'parity', d.get('parity',{}).get('parity_ok'), 'roofline', d.get('roofline'))
print('bwt_extends', d['config']['bwt_extends'], 'seeds', d['config']['seeds'])
for k in ('fused','text'):
    r=L.get(k,{}); print(k, {x:r.get(x) for x in ('pairs','wall_s','pairs_per_s','error')})
print('config5', d.get('config5'))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $repo/bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e --no-profile --config5-pairs 0 --no-dist-rehearsal > $out/r04g_bench_under_rocprof.json 2> $out/r04g_rocprof.err
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then (head -1 $f; grep ssg_k $f) > $out/r04g_kernel_stats_ssg.csv; head -12 $out/r04g_kernel_stats_ssg.csv | cut -c1-160; fi
