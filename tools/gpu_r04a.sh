#!/bin/bash
# Round 4, first GPU call (~12 GPU-minutes): what round 3 built after its GPU minutes ran out, measured.
#   1. pytest -m gpu (fused frames as mapped segments, several-thread FASTQ reader, sort-written BAI -- green on the emulation only so far)
#   2. the bench line as the driver runs it (literal metric on 8 M pairs with the stage timers: index load split, samblaster's main thread,
#      the sort's input / write / index split)
#   3. the same script leg with the frame payloads back on the pipes (SSG_FUSED_SHM=0): what the segments are worth on this box
#   3b. the same leg with two device calls in flight per GPU (SSG_BWA_INFLIGHT=2, lanes)
#   4. config 3 soak: 40 M pairs through the script on one GPU (rate, peak RSS, spills, flagstat-level invariants)
#   5. kernel-trace stats of the device step for profiles/r04_*
#   6. the reproducer for DESIGN.md section 9's open item (run `tools/dbg/smem_variants.sh build` HERE before the gpurun call: the variant
#      libraries travel with the snapshot)
out=$PWD/gpurun_out; mkdir -p $out; repo=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > $out/r04a_pytest_gpu.log 2>&1; tail -3 $out/r04a_pytest_gpu.log
SSG_TEST_INFLIGHT=1 timeout 400 python -m pytest tests/test_zz_twins_gpu.py -m gpu -q -k in_flight > $out/r04a_pytest_inflight.log 2>&1; tail -2 $out/r04a_pytest_inflight.log   # lanes: two device calls in flight per GPU
timeout 900 python bench.py --steps 5 --warmup 2 > $out/r04a_bench.json 2> $out/r04a_bench.err; tail -4 $out/r04a_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04a_bench.json')); L=d.get('literal',{})
print('ms/step', round(d['ms_per_step'],1), 'value', d.get('value'), 'parity', d.get('parity',{}).get('parity_ok'))
for k in ('fused','text'):
    r=L.get(k,{}); print(k, {x:r.get(x) for x in ('pairs','wall_s','pairs_per_s','error')})
    for l in r.get('stage_log',[]): print('   ', l)
e=d.get('e2e',{}); print('plugin path (text hand-off):', {k:e.get(k) for k in ('pairs','pairs_per_s','index_load_s','pairs_per_s_gz_input','gz_input_pairs')})   # gz: decoder of our own, several threads on the one stream + several parse threads
print('config5', d.get('config5')); print('dist', d.get('dist_rehearsal')); print('cpu script', d.get('cpu_baseline',{}).get('script'))
PY
SSG_BENCH_CONFIG_EXTRA="export SSG_FUSED_SHM=0" timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 2000 --config5-pairs 0 --cpu-script-pairs 0 --no-dist-rehearsal --no-profile > $out/r04a_bench_pipes.json 2> $out/r04a_bench_pipes.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04a_bench_pipes.json')); r=d.get('literal',{}).get('fused',{})
print('payloads on the pipes:', {x:r.get(x) for x in ('pairs','wall_s','pairs_per_s','error')})
for l in r.get('stage_log',[]): print('   ', l)
PY
SSG_BENCH_CONFIG_EXTRA="export SSG_BWA_INFLIGHT=2" timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 2000 --config5-pairs 0 --cpu-script-pairs 0 --no-dist-rehearsal --no-profile > $out/r04a_bench_inflight2.json 2> $out/r04a_bench_inflight2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04a_bench_inflight2.json')); r=d.get('literal',{}).get('fused',{})
print('two calls in flight per GPU:', {x:r.get(x) for x in ('pairs','wall_s','pairs_per_s','error')}, 'sample BAMs equal oracle:', d.get('literal',{}).get('sample_bams_equal_oracle'))
for l in r.get('stage_log',[])[:4]: print('   ', l)
PY
timeout 900 python tools/soak.py --pairs 40000000 > $out/r04a_soak.json 2> $out/r04a_soak.err; tail -3 $out/r04a_soak.err; head -c 1500 $out/r04a_soak.json
timeout 600 python tools/soak.py --pairs 20000000 --mem 12 > $out/r04a_soak_spill.json 2> $out/r04a_soak_spill.err; head -c 1500 $out/r04a_soak_spill.json   # -M 12: sorted runs + the merge per range of the genome
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $repo/bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e --no-profile --config5-pairs 0 --no-dist-rehearsal > $out/r04a_bench_under_rocprof.json 2> $out/r04a_rocprof.err
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then (head -1 $f; grep ssg_k $f) > $out/r04a_kernel_stats_ssg.csv; head -8 $out/r04a_kernel_stats_ssg.csv; fi
cd $repo && bash tools/dbg/smem_variants.sh run
