#!/bin/bash
# First GPU call of the next round (≈ 4 min): the measurements DESIGN.md section 10 lists as open.
#   1. SMEM counters of the tune build (tools/dbg/phase.py): cycles in the state machine vs at the extension site, lane readiness
#   2. A/B of the nested-loop SMEM kernel (SSG_SMEM_KERNEL=lane), now on the one-round-trip extension
#   3. device-call size of `bwa mem` (SSG_BWA_CALL_PAIRS) with its per-kernel profile (SSG_BWA_PROF) through the bench's e2e leg
#   4. TCC hit / miss counters of the FM-index kernels (separate --pmc pass, kernel trace only)
out=$PWD/gpurun_out; mkdir -p $out; repo=$PWD
timeout 300 python tools/dbg/phase.py > $out/next_phase.txt 2>&1; tail -8 $out/next_phase.txt
SSG_SMEM_KERNEL=lane timeout 240 python bench.py --steps 3 --warmup 1 --no-e2e --cpu-sample 2000 > $out/next_lane.json 2> $out/next_lane.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/next_lane.json')); k=d['roofline']['kernels_ms_per_step']
print('lane kernel: ms/step', round(d['ms_per_step'],1), d.get('parity',{}).get('parity_ok'), {x:k[x] for x in list(k)[:4]})
PY
for cp in 524288 1048576; do
  SSG_BWA_CALL_PAIRS=$cp SSG_E2E_STAGE_LOG=$out/next_e2e_$cp.log timeout 600 python bench.py --steps 2 --warmup 1 > $out/next_e2e_$cp.json 2> $out/next_e2e_$cp.err
  python - $cp <<'PY'
import json,sys
d=json.load(open('gpurun_out/next_e2e_%s.json' % sys.argv[1])); e=d.get('e2e',{})
print('call pairs', sys.argv[1], {k:e.get(k) for k in ('reads_to_sam_s','pairs_per_s','bwa_stage_busy')}, e.get('speedseq_align_script'))
PY
  grep "\[bwa\] kernel" $out/next_e2e_$cp.log | head -8
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d /tmp/pmc_tcc -o pmc -- python $repo/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-profile > $out/next_pmc_tcc.log 2>&1
f=$(find /tmp/pmc_tcc -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then (head -1 $f; grep "ssg_k_smem\|ssg_k_sal" $f) > $out/next_pmc_tcc.csv; wc -l $out/next_pmc_tcc.csv; fi
