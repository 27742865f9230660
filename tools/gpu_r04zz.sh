#!/bin/bash
# Round 4, last GPU seconds: one pipeline against two pipelines side by side on the ONE GPU (rank mode), 3 M pairs: what the second pipeline's host stages buy when the device is shared.
out=$PWD/gpurun_out; mkdir -p $out
timeout 58 python tools/dbg/literal_ab.py --pairs 3000000 --ref-mbp 800 --no-warmup --out $out/r04zz_ranks_one_gpu.json one:t=16 two:t=16:ranks=2 > $out/r04zz_ranks_one_gpu.log 2>&1
grep -E "config|speedseq-ranks" $out/r04zz_ranks_one_gpu.log | cut -c1-200
