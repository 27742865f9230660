#!/bin/bash
# round 2, pass o: the whole GPU suite, then the default bench (parity gate on 20000 pairs, cpu baseline, plugin-path leg)
out=$PWD/gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $out/r02o_pytest_gpu.log
timeout 900 python bench.py --steps 5 --warmup 2 > $out/r02o_bench.json 2> $out/r02o_bench.err || tail -5 $out/r02o_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02o_bench.json'))
print('ms/step', round(d['ms_per_step'],1), d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['parity']['parity_ok'], d['cpu_baseline'], d.get('e2e'))
PY
