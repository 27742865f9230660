#!/bin/bash
# Round 3, fourth GPU call: (1) the k-mer table's self-check on the GPU (table vs forward extension), (2) GPU suite with the table off (default) and on,
# (3) the bench with its legs at reduced sizes (table off): where the time goes, e2e and literal numbers.
out=$PWD/gpurun_out; mkdir -p $out
python - <<'PY'
import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'tools')
import simreads
simreads.write_fastq('/tmp/t.fq', simreads.simulate(simreads.read_fasta('tests/golden/chr20_slice.fa'), 2000, seed=9))
PY
for k in 7 9; do SSG_KTAB_K=$k SSG_KTAB_VERIFY=1 timeout 120 bin/bwa mem -p tests/golden/chr20_slice.fa /tmp/t.fq 2>&1 >/tmp/k$k.sam | grep "k-mer"; done
timeout 120 bin/bwa mem -p tests/golden/chr20_slice.fa /tmp/t.fq 2>/dev/null > /tmp/k0.sam; for k in 7 9; do echo "K=$k SAM lines differing from K=0: $(diff <(grep -v '^@PG' /tmp/k0.sam) <(grep -v '^@PG' /tmp/k$k.sam) | grep -c '^<')"; done
timeout 600 python -m pytest tests -m gpu -x -q > $out/r03d_pytest_gpu.log 2>&1; tail -2 $out/r03d_pytest_gpu.log
SSG_KTAB_K=7 timeout 300 python -m pytest tests -m gpu -q -k "smem or cli_gpu or align1" > $out/r03d_pytest_ktab.log 2>&1; tail -4 $out/r03d_pytest_ktab.log
timeout 460 python bench.py --steps 3 --warmup 1 --cpu-sample 20000 --e2e-pairs 4000000 --script-pairs 4000000 --cpu-script-pairs 50000 --partial $out/r03d_partial.json > $out/r03d_bench.json 2> $out/r03d_bench.err; echo "bench rc=$?"
grep "^\[bench" $out/r03d_bench.err | tail -30
python - <<'PY'
import json,os
p='gpurun_out/r03d_bench.json'
d=json.load(open(p if os.path.getsize(p) else 'gpurun_out/r03d_partial.json'))
print('value', d['value'], 'ms/step', round(d['ms_per_step'],1), 'parity', json.dumps(d.get('parity',{}))[:500])
k=d.get('roofline',{}).get('kernels_ms_per_step',{}); print({x:k[x] for x in list(k)[:10]}, 'frac', d.get('roofline',{}).get('frac'))
print('cpu', json.dumps(d.get('cpu_baseline',{}))[:900])
e=d.get('e2e',{}); print('e2e', {k:e.get(k) for k in ('index_load_s','reads_to_sam_s','pairs_per_s','bwa_stage_busy','pairs_per_s_gz_input','sample_streams_identical','error')})
print('literal', json.dumps(d.get('literal',{}),indent=1)[:3500])
PY
SSG_KTAB_K=13 SSG_KTAB_VERIFY=1 timeout 150 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --partial $out/r03d_k13.json 2>&1 | grep -E "k-mer|bench " | tail -4
