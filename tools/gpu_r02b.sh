#!/bin/bash
# round 2, pass b: full -m gpu suite, then the bench with the parity gate and the plugin-path (e2e) leg
out=$PWD/gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/r02b_pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"
tail -5 $out/r02b_pytest_gpu.log
timeout 1200 python bench.py --steps 3 --warmup 1 > $out/r02b_bench.json 2> $out/r02b_bench.err; echo "bench rc=$?"
tail -c 3500 $out/r02b_bench.json; grep -v "ssg index" $out/r02b_bench.err | tail -20
