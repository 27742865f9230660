#!/bin/bash
# kernel trace of the literal pipeline (bench.py's script leg: `speedseq align` on bin/bwa, bin/samblaster, bin/sambamba) at a reduced size: which process keeps the device busy, with what, and how much of the wall it idles
tag=${1:-bwa_trace}; pairs=${2:-3000000}
out=$PWD/gpurun_out; mkdir -p $out
B="python $PWD/bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-e2e --no-profile --config5-pairs 0 --no-dist-rehearsal --script-pairs $pairs --cpu-script-pairs 0"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -o $tag -- $B > $out/${tag}_bench.json 2> $out/${tag}_bench.err)
python tools/dbg/trace_busy.py /tmp/prof_$tag 22 > $out/${tag}_busy.txt 2>&1; head -120 $out/${tag}_busy.txt
python - $out/${tag}_bench.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); L = d.get('literal', {}).get('fused', {})
    print({k: L.get(k) for k in ('pairs', 'wall_s', 'pairs_per_s')}); print("\n".join(L.get('stage_log', [])[:6]))
except Exception as e: print("no bench line", e)
PY
