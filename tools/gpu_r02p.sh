#!/bin/bash
# round 2, pass p: realign through the reference script on the GPU; bench with the 4 M-pair plugin-path leg + stage log of bwa
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_realign.py -m gpu -x -q 2>&1 | tail -3
SSG_E2E_STAGE_LOG=$out/r02p_bwa_stages.log timeout 1200 python bench.py --steps 3 --warmup 1 > $out/r02p_bench.json 2> $out/r02p_bench.err || tail -5 $out/r02p_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02p_bench.json'))
print('ms/step', round(d['ms_per_step'],1), d['value'], d['parity']['parity_ok'], d.get('e2e'))
PY
grep -c "stage" $out/r02p_bwa_stages.log; grep "wall:\|busy" $out/r02p_bwa_stages.log
