#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats, and separate PMC passes (FETCH_SIZE / WRITE_SIZE) for
# the ssg_k_* kernels plus a calibration pass on the random-gather probe.  Usage: tools/profile_round.sh r01d
tag=${1:-rXX}; out=$PWD/gpurun_out; mkdir -p $out
timeout 280 python bench.py > $out/bench_$tag.json 2> $out/bench_$tag.err
cd /tmp && export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python /root/repo/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-profile > $out/${tag}_rocprof_bench.log 2>&1
find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$c -o pmc -- python /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-profile > $out/${tag}_pmc_$c.log 2>&1
  f=$(find /tmp/pmc_${tag}_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep "ssg_k_" $f) > $out/${tag}_pmc_$c.csv; fi
done
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_${tag}_probe -o pmc -- /root/repo/tools/dbg/gather_probe > $out/${tag}_probe.log 2>&1
f=$(find /tmp/pmc_${tag}_probe -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then (head -1 $f; grep "probe" $f) > $out/${tag}_pmc_probe.csv; fi
ls -la $out | tail -12
