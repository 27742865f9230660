#!/bin/bash
# One GPU call, assembled from steps (replaces the one-shot tools/gpu_rNN?.sh scripts of rounds 2-4).
#   gpurun --timeout S -- 'bash tools/gpu_run.sh TAG STEP [STEP...]'
# Every step writes gpurun_out/TAG_<what> and prints a few summary lines; a step that fails does not stop the later ones.
#   suite[:K_EXPR]            pytest -m gpu (optionally -k K_EXPR)
#   bench[:ARGS]              bench.py --steps 5 --warmup 2 ARGS  (ARGS with '+' for spaces, e.g. bench:--cpu-script-pairs+0)
#   ab:KERNELS:CFG[;CFG...]   tools/smem_ab.py --kernels KERNELS CFG...   (A/B of variants inside the device step)
#   ab250:KERNELS:CFG[;...]   the same at 2x250, 200 k pairs
#   profile                   tools/profile_round.sh TAG (kernel stats + PMC passes + probes) and tools/pmc_summarize.py
#   profile250                the same on the 2x250 workload (BASELINE.json configs[4], 200 k pairs), no probes: TAG_cfg5_*
#   stats                     only the rocprofv3 --kernel-trace --stats pass of the device step
#   timeline[:ARGS]           kernel trace of the device step -> which kernels run alone, in what order (TAG_timeline.txt)
#   soak:PAIRS[:MEM_GB]       tools/soak.py --stream --pairs PAIRS [--mem MEM_GB]   (the FASTQ goes through a FIFO, never a file)
#   pin                       write the kernel ISA pin if the suite log and the bench parity of this TAG are green
#   sh:CMD                    any other command (with '+' for spaces)
tag=$1; shift
out=$PWD/gpurun_out; mkdir -p $out
for step in "$@"; do
  what=${step%%:*}; arg=""; [ "$step" != "$what" ] && arg=${step#*:}
  t0=$(date +%s)
  case $what in
  suite)
    if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "${arg//+/ }" > $out/${tag}_pytest_gpu.log 2>&1
    else timeout 1500 python -m pytest tests -m gpu -x -q > $out/${tag}_pytest_gpu.log 2>&1; fi
    tail -4 $out/${tag}_pytest_gpu.log | cut -c1-300 ;;
  bench)
    timeout 1200 python bench.py --steps 5 --warmup 2 ${arg//+/ } > $out/${tag}_bench.json 2> $out/${tag}_bench.err; tail -2 $out/${tag}_bench.err | cut -c1-300
    python - $out/${tag}_bench.json <<'PY'
import json, sys
try: d = json.load(open(sys.argv[1]))
except Exception as e: print("no bench line:", e); sys.exit(0)
r = d.get('roofline', {}); L = d.get('literal', {})
print('ms/step', round(d['ms_per_step'], 1), 'value', d.get('value'), 'literal', (d.get('value_literal') or {}).get('value'), 'parity', d.get('parity', {}).get('parity_ok'), 'bwt_extends', d['config'].get('bwt_extends'))
print('roofline', {k: r.get(k) for k in ('kernel', 'achieved', 'frac', 'traffic', 'ms_per_launch', 'largest_kernel')})
print('kernels', list(r.get('kernels_ms_per_step', {}).items())[:16])
for k in ('fused', 'text'):
    x = L.get(k, {}); print(k, {y: x.get(y) for y in ('pairs', 'wall_s', 'pairs_per_s', 'error')})
print('cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'))
print('config5', {k: d.get('config5', {}).get(k) for k in ('ms_per_step', 'pairs_per_s', 'parity_ok')})
PY
    ;;
  ab|ab250)
    kern=${arg%%:*}; cfgs=${arg#*:}
    extra=""; [ $what = ab250 ] && extra="--read-len 250 --pairs 200000"
    timeout 900 python tools/smem_ab.py $extra --kernels "$kern" --out $out/${tag}_${what}.json ${cfgs//;/ } > $out/${tag}_${what}.log 2>&1
    grep -E "\"config\"|summary counts|Error|error" $out/${tag}_${what}.log | cut -c1-600 ;;
  profile)
    bash tools/profile_round.sh $tag > $out/${tag}_profile_round.log 2>&1; tail -5 $out/${tag}_profile_round.log
    python tools/pmc_summarize.py $tag $out $out 2>&1 | tail -20 ;;
  profile250)
    BENCH_EXTRA="--read-len 250 --pairs 200000" PROFILE_PROBES=0 bash tools/profile_round.sh ${tag}_cfg5 > $out/${tag}_cfg5_profile_round.log 2>&1; tail -3 $out/${tag}_cfg5_profile_round.log
    python tools/pmc_summarize.py ${tag}_cfg5 $out $out 2>&1 | tail -20 ;;
  stats)
    B="python $PWD/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-profile --config5-pairs 0 --no-dist-rehearsal --script-pairs 0 --cpu-script-pairs 0"
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- $B > $out/${tag}_bench_under_rocprof.log 2>&1)
    find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
    grep ssg_k_ $out/${tag}_kernel_stats.csv | cut -c1-60,200- | head -30 ;;
  timeline)   # the last device step of a traced run: which kernels run alone (tools/dbg/kernel_timeline.py)
    B="python $PWD/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-profile --config5-pairs 0 --no-dist-rehearsal --script-pairs 0 --cpu-script-pairs 0 ${arg//+/ }"
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o $tag -- $B > $out/${tag}_bench_under_trace.log 2>&1)
    f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
    python tools/dbg/kernel_timeline.py $f --json $out/${tag}_timeline.json > $out/${tag}_timeline.txt 2>&1; head -60 $out/${tag}_timeline.txt | cut -c1-200 ;;
  soak)
    pairs=${arg%%:*}; mem=""; [ "$arg" != "$pairs" ] && mem="--mem ${arg#*:}"
    timeout 3000 python tools/soak.py ${SOAK_MODE:---stream} --limit 2400 --pairs $pairs $mem > $out/${tag}_soak_$pairs.json 2> $out/${tag}_soak_$pairs.err; tail -3 $out/${tag}_soak_$pairs.err | cut -c1-300
    python -c "import json,sys; d=json.load(open('$out/${tag}_soak_$pairs.json')); print({k:v for k,v in d.items() if k not in ('stage_log','what')})" ;;
  pin)
    python - $out/${tag}_bench.json $out/${tag}_pytest_gpu.log $out/${tag}_kernel_isa.sha256 <<'PY'
import json, subprocess, sys
try:
    ok = json.load(open(sys.argv[1])).get('parity', {}).get('parity_ok'); log = open(sys.argv[2]).read()
except Exception as e: print("pin: nothing to go by:", e); sys.exit(0)
if ok and ' passed' in log and 'failed' not in log and 'error' not in log.lower():
    print(subprocess.run([sys.executable, 'tools/isa_pin.py', '--write', '--golden', sys.argv[3]], capture_output=True, text=True).stdout)
else: print("pin: suite or parity gate not green, no pin written")
PY
    ;;
  sh) timeout 1500 bash -c "${arg//+/ }" 2>&1 | tail -20 | cut -c1-300 ;;
  *) echo "gpu_run.sh: unknown step $step" ;;
  esac
  echo "[gpu_run] $step: $(( $(date +%s) - t0 )) s"
done
