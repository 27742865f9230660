#!/bin/bash
# Round 4, 19th GPU call: the sort's spill path after its rewrite (stretch-wise merge through the device, index written by the sort): 24 M pairs with -M 8
# (budget 3.6 GB of records per run: ~4 runs, ~13 stretches), stage log kept.
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python tools/soak.py --pairs 24000000 --mem 8 > $out/r04s_soak_24M_spill.json 2> $out/r04s_soak_24M_spill.err; tail -3 $out/r04s_soak_24M_spill.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04s_soak_24M_spill.json'))
print({k:v for k,v in d.items() if k not in ('stage_log','what')})
for l in d.get('stage_log',[]):
    if 'merge' in l or 'runs' in l or 'wall' in l or 'input thread' in l: print('   ', l[:400])
PY
