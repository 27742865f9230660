#!/bin/bash
# round 2, final pass: whole GPU suite, default bench (parity gate, cpu baseline, plugin-path leg), then the round profile on the same tree
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $out/r02s_pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > $out/r02s_bench.json 2> $out/r02s_bench.err || tail -5 $out/r02s_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02s_bench.json'))
print('ms/step', round(d['ms_per_step'],1), d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['parity']['parity_ok'], d['config']['seeds'], d.get('e2e'))
PY
bash tools/profile_round.sh r02 > $out/r02s_profile.log 2>&1
tail -3 $out/r02s_profile.log
