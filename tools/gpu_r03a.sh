#!/bin/bash
# Round 3, first GPU call: state of the tree after the per-device runtime + streamed index load, and the measurements the round plans from.
#   0. the box (cores, memory)   1. pytest -m gpu   2. bench line with the plugin-path leg (index load timing, stage log)
#   3. SMEM phase counters of the tune build     4. TCC hit / miss counters of the FM-index kernels
out=$PWD/gpurun_out; mkdir -p $out; repo=$PWD
(nproc; free -g | head -2; lscpu | grep -E "Model name|Socket|NUMA node\(s\)"; df -h /dev/shm | tail -1) > $out/r03a_box.txt 2>&1; cat $out/r03a_box.txt
timeout 900 python -m pytest tests -m gpu -x -q > $out/r03a_pytest_gpu.log 2>&1; tail -3 $out/r03a_pytest_gpu.log
SSG_DEBUG=1 timeout 120 bin/bwa mem 2>&1 | head -2
SSG_E2E_STAGE_LOG=$out/r03a_e2e_stage.log timeout 900 python bench.py --steps 3 --warmup 1 > $out/r03a_bench.json 2> $out/r03a_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03a_bench.json')); e=d.get('e2e',{})
print('ms/step', round(d['ms_per_step'],1), 'parity', d.get('parity',{}).get('parity_ok'), 'cpu', d.get('cpu_baseline',{}).get('value'))
print({k:e.get(k) for k in ('index_load_s','reads_to_sam_s','pairs_per_s','bwa_stage_busy','pairs_per_s_gz_input')}, e.get('speedseq_align_script'))
k=d['roofline']['kernels_ms_per_step']; print({x:k[x] for x in list(k)[:12]})
PY
grep -i "index load" $out/r03a_e2e_stage.log | head -3
timeout 300 python tools/dbg/phase.py > $out/r03a_phase.txt 2>&1; tail -4 $out/r03a_phase.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d /tmp/pmc_tcc -o pmc -- python $repo/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-profile > $out/r03a_pmc_tcc.log 2>&1
f=$(find /tmp/pmc_tcc -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then (head -1 $f; grep "ssg_k_smem\|ssg_k_sal" $f) > $out/r03a_pmc_tcc.csv; wc -l $out/r03a_pmc_tcc.csv; head -3 $out/r03a_pmc_tcc.csv; fi
