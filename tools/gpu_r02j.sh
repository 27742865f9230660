#!/bin/bash
# round 2, pass j: kernel timeline of one step (idle gaps between stages)
out=$PWD/gpurun_out; mkdir -p $out; repo=$PWD
B="python $repo/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-profile"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- $B > $out/r02j_tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python $repo/tools/timeline.py $f $out/r02j_timeline.txt | head -120
