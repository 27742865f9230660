#!/bin/bash
# Round 4, tenth GPU call: BGZF deflate on the device: parity (zlib inflates every block; the reference's samtools reads the sorted BAM), rate, the literal leg.
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_bgzf_device.py tests/test_sambamba.py tests/test_fused.py -m gpu -x -q > $out/r04j_pytest.log 2>&1; tail -3 $out/r04j_pytest.log
timeout 300 python tools/dbg/bgzf_bench.py 1024 2>&1 | tee $out/r04j_bgzf_bench.log | tail -6
timeout 200 python tools/dbg/host_probe.py 2>&1 | tee $out/r04j_host_probe.log | tail -12
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 2000 --config5-pairs 0 --cpu-script-pairs 0 --no-dist-rehearsal --no-profile > $out/r04j_bench_literal.json 2> $out/r04j_bench_literal.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04j_bench_literal.json')); L=d.get('literal',{})
print('sample BAMs equal oracle:', L.get('sample_bams_equal_oracle'))
for k in ('fused','text'):
    x=L.get(k,{}); print(k, {y:x.get(y) for y in ('pairs','wall_s','pairs_per_s','error','bam_bytes')})
    for l in x.get('stage_log',[]):
        if 'sort' in l or 'bwa' in l: print('   ', l[:400])
PY
