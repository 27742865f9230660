#!/bin/bash
# Round 4, 13th GPU call: device deflate with the region's own table: parity, rate and size; the literal leg (BAM sizes against the zlib-written ones).
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_bgzf_device.py tests/test_sambamba.py -m gpu -x -q > $out/r04m_pytest.log 2>&1; tail -2 $out/r04m_pytest.log
timeout 300 python tools/dbg/bgzf_bench.py 1024 2>&1 | tee $out/r04m_bgzf_bench.log | tail -5
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 2000 --config5-pairs 0 --cpu-script-pairs 0 --no-dist-rehearsal --no-profile > $out/r04m_bench_literal.json 2> $out/r04m_bench_literal.err
SSG_BENCH_CONFIG_EXTRA="export SSG_BGZF_DEVICE=0" timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 2000 --config5-pairs 0 --cpu-script-pairs 0 --no-dist-rehearsal --no-profile > $out/r04m_bench_literal_zlib.json 2> $out/r04m_bench_literal_zlib.err
python - <<'PY'
import json
for f in ('r04m_bench_literal.json','r04m_bench_literal_zlib.json'):
    d=json.load(open('gpurun_out/'+f)); L=d.get('literal',{})
    x=L.get('fused',{}); print(f, 'equal oracle:', L.get('sample_bams_equal_oracle'), {y:x.get(y) for y in ('pairs','wall_s','pairs_per_s','error','bam_bytes')})
    for l in x.get('stage_log',[]):
        if '78867' in l or '16074991' in l: print('   ', l[:330])
PY
