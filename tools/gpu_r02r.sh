#!/bin/bash
# round 2, pass r: host pipeline (parser, block-sharing batches, formatter): hotpath at 2x150 / 2x250, CLI parity on the GPU, the full bench with the plugin-path leg
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_cli.py tests/test_speedseq_script.py -m gpu -x -q -k "hotpath or cli or script" 2>&1 | tail -3
SSG_E2E_STAGE_LOG=$out/r02r_bwa_stages.log timeout 1200 python bench.py --steps 3 --warmup 1 > $out/r02r_bench.json 2> $out/r02r_bench.err || tail -5 $out/r02r_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02r_bench.json'))
print('ms/step', round(d['ms_per_step'],1), d['value'], d['parity']['parity_ok'], d.get('e2e'))
PY
grep "wall:\|busy" $out/r02r_bwa_stages.log
