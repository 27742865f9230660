#!/bin/bash
# Round 4, 17th GPU call: 80 M pairs through the reference's script with -M 48: the sort spills (sorted runs in segments per range of the genome, merge per
# range), samblaster's duplicate table doubles, bwa mem makes the denser suffix-array copy after 32 M pairs -- the paths that had only run under emulation.
out=$PWD/gpurun_out; mkdir -p $out
timeout 2000 python tools/soak.py --pairs 80000000 --mem 48 > $out/r04q_soak_80M_spill.json 2> $out/r04q_soak_80M_spill.err; tail -3 $out/r04q_soak_80M_spill.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04q_soak_80M_spill.json'))
print({k:v for k,v in d.items() if k not in ('stage_log','what')})
for l in d.get('stage_log',[]):
    print('   ', l[:300])
PY
