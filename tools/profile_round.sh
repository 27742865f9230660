#!/bin/bash
# Round profile on the GPU box (usage: tools/profile_round.sh r02): bench line, rocprofv3 kernel stats, separate PMC passes
# (HBM traffic: FETCH_SIZE / WRITE_SIZE; SQ issue counters), the two roofline probes with a FETCH_SIZE calibration pass.
# Counters are collected with --kernel-trace only (no sys/hip/hsa trace domains).  Summaries: tools/pmc_summarize.py.
# BENCH_EXTRA selects another workload of the same step (BASELINE.json configs[4]: BENCH_EXTRA="--read-len 250 --pairs 200000");
# PROFILE_PROBES=0 skips the roofline probes and their calibration passes (they do not depend on the workload).
tag=${1:-rXX}; out=$PWD/gpurun_out; mkdir -p $out
B="python $PWD/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-profile --config5-pairs 0 --no-dist-rehearsal --script-pairs 0 --cpu-script-pairs 0 $BENCH_EXTRA"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- $B > $out/${tag}_bench_under_rocprof.log 2>&1
find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$n -o pmc -- $B > $out/${tag}_pmc_$n.log 2>&1
  f=$(find /tmp/pmc_${tag}_$n -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep "ssg_k_" $f) > $out/${tag}_pmc_$n.csv; fi
done
if [ "${PROFILE_PROBES:-1}" = 0 ]; then ls -la $out | tail -20; grep -c ssg_k $out/${tag}_pmc_*.csv; exit 0; fi
$PWD/../repo/tools/dbg/gather_probe > $out/${tag}_gather_probe.txt 2>&1 || /root/repo/tools/dbg/gather_probe > $out/${tag}_gather_probe.txt 2>&1
/root/repo/tools/dbg/valu_probe > $out/${tag}_valu_probe.txt 2>&1
/root/repo/tools/dbg/libm_probe > $out/${tag}_libm_probe.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_${tag}_probe -o pmc -- /root/repo/tools/dbg/gather_probe > $out/${tag}_probe_under_pmc.log 2>&1
f=$(find /tmp/pmc_${tag}_probe -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then (head -1 $f; grep "probe\|stream_" $f) > $out/${tag}_pmc_probe_calibration.csv; fi
# ... and the write counter on the probe's streaming launches (a pass of its own: FETCH_SIZE and WRITE_SIZE do not fit one pass)
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_${tag}_probe_w -o pmc -- /root/repo/tools/dbg/gather_probe > $out/${tag}_probe_under_pmc_w.log 2>&1
f=$(find /tmp/pmc_${tag}_probe_w -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then (head -1 $f; grep "stream_" $f) > $out/${tag}_pmc_probe_calibration_write.csv; fi
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|Counter).*(TCC_HIT|TCC_MISS|TCC_REQ|SQ_INSTS_VALU|FETCH_SIZE)" | head -20 > $out/${tag}_counter_names.txt
ls -la $out | tail -20
cat $out/${tag}_libm_probe.txt; grep -c ssg_k $out/${tag}_pmc_*.csv
