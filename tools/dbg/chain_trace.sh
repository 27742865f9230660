#!/bin/bash
# kernel timeline of the chaining stage (rocprofv3 --kernel-trace of a short device-step run), then the chain A/B
tag=${1:-chain_trace}; shift
out=$PWD/gpurun_out; mkdir -p $out
B="python $PWD/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-profile --config5-pairs 0 --no-dist-rehearsal --script-pairs 0 --cpu-script-pairs 0"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -o $tag -- $B > $out/${tag}_bench_under_rocprof.log 2>&1)
f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python tools/dbg/stage_timeline.py $f > $out/${tag}_timeline.txt 2>&1; grep -v "fillBuffer\|trampoline\|lookback" $out/${tag}_timeline.txt | head -40
python tools/dbg/step_gaps.py $f 30 > $out/${tag}_step_gaps.txt 2>&1; head -34 $out/${tag}_step_gaps.txt
bash tools/dbg/chain_ab.sh ${tag}_ab "$@"
