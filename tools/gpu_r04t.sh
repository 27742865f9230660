#!/bin/bash
# Round 4, 20th GPU call: the literal metric against the host's thread counts.  The box shows 256 hardware threads and allows 16 CPUs (cgroup cpu.max):
# pools sized from the former get throttled as a group by the latter.
out=$PWD/gpurun_out; mkdir -p $out
timeout 560 python tools/dbg/literal_ab.py --out $out/r04t_literal_ab.json \
  now:t=32:SSG_SORT_THREADS=128 \
  quota:t=32:SSG_SORT_THREADS=20:SSG_FMT_THREADS=12:SSG_FASTQ_THREADS=3:SSG_SBL_THREADS=8 \
  quota3:t=32:SSG_SORT_THREADS=20:SSG_FMT_THREADS=12:SSG_FASTQ_THREADS=3:SSG_SBL_THREADS=8:SSG_BWA_INFLIGHT=3 \
  t16:t=16:SSG_SORT_THREADS=16:SSG_FMT_THREADS=8:SSG_FASTQ_THREADS=2:SSG_SBL_THREADS=6 \
  > $out/r04t_literal_ab.log 2>&1
grep -E "config|quota" $out/r04t_literal_ab.log | cut -c1-240
