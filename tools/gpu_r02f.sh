#!/bin/bash
# round 2, pass f: whole -m gpu suite, headline bench (parity gate, validators, e2e), single-rank rehearsal of the N>1 step, libm probe
out=$PWD/gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/r02f_pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -4 $out/r02f_pytest_gpu.log
timeout 1200 python bench.py --steps 5 --warmup 2 > $out/r02f_bench.json 2> $out/r02f_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02f_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['random64B']['frac'], d['roofline']['sw'])
print(json.dumps(d.get('e2e'), indent=1)); print(json.dumps(d.get('parity'), indent=1)); print(d.get('cpu_baseline'))
PY
grep -v "ssg index" $out/r02f_bench.err | tail -5
SSG_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-e2e --cpu-sample 0 > $out/r02f_bench_forcedist.json 2> $out/r02f_forcedist.err; echo "forcedist rc=$?"; tail -c 700 $out/r02f_bench_forcedist.json; tail -3 $out/r02f_forcedist.err
tools/dbg/libm_probe > $out/r02f_libm_probe.txt; cat $out/r02f_libm_probe.txt
