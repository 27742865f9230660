#!/bin/bash
# first GPU pass of round 2: index builder tests, small bench (flow check), headline-size bench
out=$PWD/gpurun_out; mkdir -p $out
export SSG_INDEX_VERBOSE=1
timeout 600 python -m pytest tests/test_index_build.py -m gpu -x -q > $out/r02a_pytest_index.log 2>&1; echo "pytest index rc=$?"
tail -3 $out/r02a_pytest_index.log
timeout 600 python bench.py --ref-mbp 100 --pairs 100000 --cpu-sample 5000 --steps 2 > $out/r02a_bench_small.json 2> $out/r02a_bench_small.err; echo "bench small rc=$?"
tail -c 1500 $out/r02a_bench_small.json; tail -5 $out/r02a_bench_small.err
timeout 1500 python bench.py --steps 3 --warmup 1 > $out/r02a_bench.json 2> $out/r02a_bench.err; echo "bench full rc=$?"
tail -c 6000 $out/r02a_bench.json; tail -30 $out/r02a_bench.err
