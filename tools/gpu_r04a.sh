#!/bin/bash
# Round 4, first GPU call: baseline of the tree (suite, bench line with stage timers) + the diagnosis of DESIGN.md section 9's open item
# (variants of the table instance of the seeding kernel, each with counters and an echo of the arguments the kernel sees).
out=$PWD/gpurun_out; mkdir -p $out; repo=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > $out/r04a_pytest_gpu.log 2>&1; tail -3 $out/r04a_pytest_gpu.log
KT_KS="8" bash tools/dbg/kt_variants.sh run > /dev/null 2>&1; mv $out/kt_variants.log $out/r04a_kt_variants.log; grep -E "^==|differ;|counters|kernel sees|host has|entries differing" $out/r04a_kt_variants.log
echo "== probe"; SSGPU_LIB=$PWD/speedseq_amd/libssgpu_probe.so timeout 120 python tools/dbg/smem_dump.py 500 2>&1 | tail -2
SSG_TEST_INFLIGHT=1 timeout 400 python -m pytest tests/test_zz_twins_gpu.py -m gpu -q -k in_flight > $out/r04a_pytest_inflight.log 2>&1; tail -2 $out/r04a_pytest_inflight.log
timeout 900 python bench.py --steps 5 --warmup 2 > $out/r04a_bench.json 2> $out/r04a_bench.err; tail -4 $out/r04a_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04a_bench.json')); L=d.get('literal',{})
print('ms/step', round(d['ms_per_step'],1), 'value', d.get('value'), 'parity', d.get('parity',{}).get('parity_ok'), 'roofline', d.get('roofline',{}).get('frac'))
for k in ('fused','text'):
    r=L.get(k,{}); print(k, {x:r.get(x) for x in ('pairs','wall_s','pairs_per_s','error')})
    for l in r.get('stage_log',[]): print('   ', l)
print('config5', d.get('config5')); print('cpu script', d.get('cpu_baseline',{}).get('script'))
PY
