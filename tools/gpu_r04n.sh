#!/bin/bash
# Round 4, 14th GPU call: device deflate after the table trim (rate, size), then the round's profile: kernel stats + PMC passes + probes (tools/profile_round.sh r04).
out=$PWD/gpurun_out; mkdir -p $out
timeout 300 python -m pytest tests/test_bgzf_device.py tests/test_sambamba.py -m gpu -x -q > $out/r04n_pytest.log 2>&1; tail -1 $out/r04n_pytest.log
timeout 300 python tools/dbg/bgzf_bench.py 1024 2>&1 | tee $out/r04n_bgzf_bench.log | tail -4
bash tools/profile_round.sh r04 > $out/r04_profile_round.log 2>&1; tail -12 $out/r04_profile_round.log
