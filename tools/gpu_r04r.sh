#!/bin/bash
# Round 4, 18th GPU call: mate rescue's local Smith-Waterman on packed 16-bit cells (wv_local_pk): stage and end-to-end parity, time inside the step against the 32-bit form.
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "local or pe_sam or mate_rescue or hotpath or pe_edge or pair_wave" > $out/r04r_pytest.log 2>&1; tail -2 $out/r04r_pytest.log
timeout 600 python tools/smem_ab.py --kernels matesw --out $out/r04r_matesw_ab.json pk nopk@nopk pk2 > $out/r04r_matesw_ab.log 2>&1
grep -E "\"config\"|summary counts|Error|error" $out/r04r_matesw_ab.log | cut -c12-300
timeout 300 python bench.py --steps 2 --warmup 1 --read-len 250 --pairs 200000 --cpu-sample 5000 --no-e2e --no-profile --config5-pairs 0 --no-dist-rehearsal 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2x250:', d['ms_per_step'], d['parity'].get('parity_ok'))"
